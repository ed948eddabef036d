cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -2
for e in 4 8 4 8 1; do
  echo "== TMI_ATTN_KSPLIT_MAX=$e"
  TMI_ATTN_KSPLIT_MAX=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
