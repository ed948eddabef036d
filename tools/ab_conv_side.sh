cd $GRAFT_REPO_ROOT
python -m pytest tests/test_whisper_step_gpu.py tests/test_two_rank_gpu.py tests/test_rccl_world1_gpu.py -x -q -m gpu > gpurun_out/convside_tests.log 2>&1 || { tail -30 gpurun_out/convside_tests.log; exit 1; }
tail -2 gpurun_out/convside_tests.log
run() { tag=$1; shift; env "$@" python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('$tag', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
for i in 1 2 3; do
  run "conv main" TMI_CONV_WGRAD_SIDE=0
  run "conv side" TMI_CONV_WGRAD_SIDE=1
done
run2() { tag=$1; shift; env "$@" python bench.py --workload wav2vec2 --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('$tag', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
for i in 1 2; do
  run2 "w2v chunks4" TMI_WGRAD_CHUNKS=4
  run2 "w2v chunks6" TMI_WGRAD_CHUNKS=6
  run2 "w2v chunks12" TMI_WGRAD_CHUNKS=12
done
