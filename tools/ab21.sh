cd $GRAFT_REPO_ROOT
TMI_DEFER_WGRAD=1 python -m pytest tests/test_whisper_step_gpu.py tests/test_two_rank_gpu.py -q -x 2>&1 | tail -1
for e in 0 1 0 1; do
  echo "== TMI_DEFER_WGRAD=$e"
  TMI_DEFER_WGRAD=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
