cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w2vside
python -m pytest tests/test_wav2vec2_gpu.py tests/test_workspace_guards_gpu.py -x -q -m gpu > gpurun_out/w2vside/tests.log 2>&1 || { tail -30 gpurun_out/w2vside/tests.log; exit 1; }
tail -2 gpurun_out/w2vside/tests.log
run() { tag=$1; shift; env "$@" python bench.py --workload wav2vec2 --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('$tag', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
for i in 1 2; do
  run "old(chunks1,conv main)" TMI_WGRAD_CHUNKS=1 TMI_CONV_WGRAD_SIDE=0
  run "chunks2 conv main" TMI_WGRAD_CHUNKS=2 TMI_CONV_WGRAD_SIDE=0
  run "chunks1 conv side" TMI_WGRAD_CHUNKS=1 TMI_CONV_WGRAD_SIDE=1
  run "chunks2 conv side" TMI_WGRAD_CHUNKS=2 TMI_CONV_WGRAD_SIDE=1
  run "chunks3 conv side" TMI_WGRAD_CHUNKS=3 TMI_CONV_WGRAD_SIDE=1
  run "chunks4 conv side" TMI_WGRAD_CHUNKS=4 TMI_CONV_WGRAD_SIDE=1
done
