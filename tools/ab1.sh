set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_whisper_step_gpu.py tests/test_two_rank_gpu.py tests/test_graph_step_gpu.py tests/test_checkpoint_gpu.py tests/test_train_loops_gpu.py tests/test_wav2vec2_gpu.py -x -q 2>&1 | tail -5
for cfg in "0 48" "1 16" "1 48" "1 128" "1 512"; do
  set -- $cfg
  echo "== ADAM_UNDER_BACKWARD=$1 blocks=$2"
  TMI_ADAM_UNDER_BACKWARD=$1 TMI_ADAM_OVERLAP_BLOCKS=$2 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "ms/step|value" | cut -c1-200
done
