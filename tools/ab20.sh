cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "eight_phase" 2>&1 | tail -1
for e in 1 0 1 0; do
  echo "== large TMI_GEMM_NO_P8_192=$e"
  TMI_GEMM_NO_P8_192=$e python bench.py --model_type large --dropout off --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
for e in 1 0; do
  echo "== medium TMI_GEMM_NO_P8_192=$e"
  TMI_GEMM_NO_P8_192=$e python bench.py --model_type medium --dropout off --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
