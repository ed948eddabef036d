"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and, for the
GEMM / attention kernels, per-grid-shape timings.  usage: prof_summary.py <dir> <steps>"""
import collections, csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
st = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(st)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.2f} ms/step over {steps:.0f} steps")
for r in rows[:22]:
    print(f"{r['Name'][:84]:84s} n/step={float(r['Calls']) / steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.3f} avg_us={float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
agg = collections.defaultdict(list)
for r in csv.DictReader(open(tr)):
    n = r["Kernel_Name"]
    if "gemm" in n or "attn" in n:
        key = (n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:44], int(r["Grid_Size_X"]) // max(1, int(r.get("Workgroup_Size_X") or 256)), r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("-- gemm/attn by grid (workgroups x, y, z)")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
    print(f"{str(k):90s} n/step={len(v) / steps:6.1f} avg={sum(v) / len(v):8.1f}us ms/step={sum(v) / 1e3 / steps:7.3f}")
