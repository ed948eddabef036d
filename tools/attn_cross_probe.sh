# usage (GPU box, repo root): bash tools/attn_cross_probe.sh  -> gpurun_out/attn_cross_probe.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/attn_cross_prof
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/attn_cross_probe.py > gpurun_out/attn_cross_probe.log 2>&1
python3 - <<'PY' | tee gpurun_out/attn_cross_probe.txt
import csv, glob
f = glob.glob("gpurun_out/attn_cross_prof/**/*kernel_trace.csv", recursive=True)[0]
R = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
CONFIGS = [(8, 100), (8, 64), (8, 128), (8, 256), (8, 512), (4, 100), (2, 100), (1, 100)]
groups, last = [], None
for r in R:
    n = r["Kernel_Name"]
    if "attn_bwd" not in n and "attn_combine" not in n:
        continue
    short = "dq" if "bwd_dq" in n else ("dkv" if "bwd_dkv" in n else "combine")
    wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * max(1, int(r["Grid_Size_Y"])) * max(1, int(r["Grid_Size_Z"]))
    key = (short, wg)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    groups.append((key, d))
# consecutive runs of 10 launches of the same (kernel, grid) per configuration
from collections import OrderedDict
seq = []
for key, d in groups:
    if seq and seq[-1][0] == key and False:
        pass
    seq.append((key, d))
# fold: configurations appear in order; each config = 10 x [dq, (combine), dkv]
i, ci = 0, 0
agg = OrderedDict()
for key, d in seq:
    agg.setdefault(key, []).append(d)
for key, ds in agg.items():
    ds = ds[2:] if len(ds) > 4 else ds
    print(f"{key[0]:8s} wgs={key[1]:6d} n={len(ds):3d} avg={sum(ds)/len(ds):7.1f} us min={min(ds):7.1f}")
PY
