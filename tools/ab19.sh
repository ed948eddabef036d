cd $GRAFT_REPO_ROOT
python -m pytest tests/test_wav2vec2_gpu.py tests/test_workspace_guards_gpu.py tests/test_two_rank_gpu.py -q -x 2>&1 | tail -2
for e in 0 1 0 1; do
  echo "== TMI_LN_EMIT=$e"
  TMI_LN_EMIT=$e python bench.py --workload wav2vec2 --steps 200 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed|host enq" | cut -c1-200
done
python tools/host_step_time_w2v.py 2>&1 | tail -1
