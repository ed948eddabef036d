cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -k "flash" 2>&1 | tail -3
for cfg in "0 3" "1 3" "1 2"; do
  set -- $cfg
  for dp in 0 0.1; do
  echo "== V2=$1 OCC=$2 dropout=$dp"
  TMI_ATTN_V2=$1 TMI_ATTN_V2_OCC=$2 ATTN_DROPOUT=$dp python tools/attn_bench.py 2>&1 | grep -E "fwd" | head -3
  done
done
