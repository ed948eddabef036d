cd $GRAFT_REPO_ROOT
python -m pytest tests/test_whisper_step_gpu.py tests/test_kernels_gpu.py tests/test_two_rank_gpu.py tests/test_full_size_properties_gpu.py -x -q 2>&1 | tail -3
for e in 0 1; do
  echo "== TMI_LN_EMIT=$e"
  TMI_LN_EMIT=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
