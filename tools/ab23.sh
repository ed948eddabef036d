cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_whisper_step_gpu.py -q -x -k "gemm or step_grad" 2>&1 | tail -1
for e in 0 1 0 1; do
  echo "== TMI_GEMM_P8_192_SHORTK=$e"
  TMI_GEMM_P8_192_SHORTK=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
