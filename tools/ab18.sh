cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_whisper_step_gpu.py -q -x -k "xent or golden or step_grad" 2>&1 | tail -2
for e in 1 0 1 0; do
  echo "== TMI_XENT_GENERIC=$e"
  TMI_XENT_GENERIC=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
