# SQ counters of the attention kernels (encoder shape): where do the wave cycles go?
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/attn_pmc
rm -rf $O; mkdir -p $O
cat > $O/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tethys_speech_amd import ops
dev = torch.device("cuda:0")
B, H, HD, T = 8, 12, 64, 1500
D = H * HD
DROP = float(os.environ.get("ATTN_DROPOUT", "0"))
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda: (torch.randn(B, T, D, device=dev, generator=g)).to(torch.bfloat16)
q, k, v, do = mk(), mk(), mk(), mk()
o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
stats = torch.empty(B, H, T, 2, device=dev); delta = torch.empty(B, H, T, device=dev)
sc = HD ** -0.5
Q = (q, 0, T * D, D); K = (k, 0, T * D, D); V = (v, 0, T * D, D); O = (o, 0, T * D, D)
for _ in range(4):
    ops.attn_fwd(Q, K, V, O, stats, B, H, T, T, 0, score_scale=sc, dropout_p=DROP, dropout_seed=77)
    ops.attn_bwd(Q, K, V, O, stats, (do, 0, T * D, D), (dq, 0, T * D, D), (dk, 0, T * D, D), (dv, 0, T * D, D), delta, B, H, T, T, 0,
                 score_scale=sc, dropout_p=DROP, dropout_seed=77)
torch.cuda.synchronize()
PY
rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -c . $O/counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  for dp in 0 0.1; do
    ATTN_DROPOUT=$dp rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${i}_$dp -- python3 $O/one.py > $O/p${i}_$dp.log 2>&1 || echo "set $i failed"
  done
done
python3 - <<'PY'
import csv, glob, collections
O = "gpurun_out/attn_pmc"
for dp in ("0", "0.1"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for f in glob.glob(f"{O}/p*_{dp}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            name = "fwd" if "attn_fwd" in k else "dq" if "attn_bwd_dq" in k else "dkv" if "attn_bwd_dkv" in k else None
            if not name: continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(name, r["Counter_Name"])] += 1
    print("dropout", dp)
    for name in ("fwd", "dq", "dkv"):
        print(" ", name, {c: round(v / max(1, cnt[(name, c)])) for c, v in sorted(agg[name].items())})
PY
