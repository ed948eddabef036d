# late Adam slices (train.ADAM_LATE): parity tests, then A/B on one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_wav2vec2_gpu.py tests/test_whisper_step_gpu.py tests/test_kernels_gpu.py tests/test_checkpoint_gpu.py tests/test_workspace_guards_gpu.py tests/test_two_rank_gpu.py -x -q -m gpu > gpurun_out/late_tests.log 2>&1 || { tail -40 gpurun_out/late_tests.log; exit 1; }
tail -2 gpurun_out/late_tests.log
run() { tag=$1; shift; env "$@" python bench.py --workload wav2vec2 --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('$tag', round(json.loads(sys.stdin.read())['ms_per_step'],3))"; }
for i in 1 2 3; do
  run "w2v late off" TMI_ADAM_LATE=0
  run "w2v late 128" TMI_ADAM_LATE=1
  run "w2v late 256" TMI_ADAM_LATE=1 TMI_ADAM_LATE_BLOCKS=256
  run "w2v late 64" TMI_ADAM_LATE=1 TMI_ADAM_LATE_BLOCKS=64
  run "w2v late full" TMI_ADAM_LATE=1 TMI_ADAM_LATE_BLOCKS=0
done
