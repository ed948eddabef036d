# Matrix-pipe utilisation of the step's kernels from hardware counters (its own rocprofv3 run: --pmc with --kernel-trace only):
# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GPU-active cycles x 1024 SIMDs), per kernel class, side stream off (every kernel alone).
# (rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs; the MfmaUtil metric of the counter list divides by their maximum)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/mfma_util
rm -rf $O; mkdir -p $O
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/p -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $O/run.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/p/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
rows = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    rows[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
def cls(k):
    if "gemm_p8" in k: return "gemm eight-phase"
    if "gemm_fast" in k or "gemm_kernel" in k: return "gemm other tiles"
    if "attn_fwd" in k: return "attention fwd"
    if "attn_bwd_dq" in k: return "attention dQ"
    if "attn_bwd_dkv" in k: return "attention dK/dV"
    return None
n = collections.Counter()
for (_, k), c in rows.items():
    name = cls(k)
    if not name or "GRBM_GUI_ACTIVE" not in c: continue
    n[name] += 1
    for kk, v in c.items(): per[name][kk] += v
print("class                 launches  MfmaUtil %   (MFMA busy / (GPU-active cycles x 1024 SIMDs); bf16 MFMA ops x 512 flop)")
tot_b = tot_a = 0.0
for name in ("gemm eight-phase", "gemm other tiles", "attention fwd", "attention dQ", "attention dK/dV"):
    c = per[name]
    if not c: continue
    util = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    tf = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (c["GRBM_GUI_ACTIVE"] / 8 / 2.4e9) / 1e12
    print(f"{name:20s} {n[name]:9d}  {util:9.1f}   ({tf:6.0f} TF/s executed at 2.4 GHz-equivalent cycles)")
    if name.startswith("gemm"):
        tot_b += c["SQ_VALU_MFMA_BUSY_CYCLES"]; tot_a += c["GRBM_GUI_ACTIVE"]
print(f"{'all GEMM kernels':20s} {'':9s}  {100.0 * tot_b / (tot_a / 8 * 1024):9.1f}")
PY
