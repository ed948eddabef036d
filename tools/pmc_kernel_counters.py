"""Per-kernel sums of arbitrary rocprofv3 --pmc counters (one pass per counter group, serial step: TMI_WGRAD_STREAM=0).
Usage: pmc_kernel_counters.py <counter_collection.csv> [<more csv>...]  -> one line per kernel family: dispatches and, per counter,
the mean per dispatch; plus the derived figures the guide names (MFMA busy share of the SIMD-cycles, LDS bank-conflict share)."""
import csv, sys, re, collections


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(gemm_p8_kernel<[^>]*>|gemm_fast_kernel<[^>]*>|attn_\w+_kernel<[^>]*>|\w+_kernel)", name)
    if m:
        return m.group(1)[:64]
    m = re.search(r"N_1\d*(\w+?_kernel)I(.*?)EEv", name)
    return (m.group(1) + "<" + m.group(2) + ">")[:64] if m else name[:64]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = family(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add((path, r["Dispatch_Id"]))
counters = sorted({c for v in agg.values() for c in v})
print("kernel family".ljust(66) + " disp " + " ".join(c[:24].rjust(24) for c in counters) + "   derived")
rows = []
for k, v in agg.items():
    n = max(1, len(ndisp[k]) // max(1, len(sys.argv) - 1))
    der = []
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CU_CYCLES" in v and v["SQ_BUSY_CU_CYCLES"] > 0:
        der.append(f"MFMA busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * v['SQ_BUSY_CU_CYCLES']):5.1f} % of busy SIMD-cycles")
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        der.append(f"LDS conflict cycles {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:5.1f} % of LDS-active")
    rows.append((-sum(v.values()), k.ljust(66) + f"{n:5d} " + " ".join(f"{v.get(c, 0) / n:24.4g}" for c in counters) + "   " + "; ".join(der)))
for _, line in sorted(rows)[:28]:
    print(line)
