set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_whisper_step_gpu.py tests/test_two_rank_gpu.py tests/test_graph_step_gpu.py tests/test_checkpoint_gpu.py -x -q 2>&1 | tail -4
python bench.py --steps 100 --warmup 5 --no-cpu-baseline 2>&1 | grep -E "timed|class" | cut -c1-2500
