set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_wav2vec2_gpu.py tests/test_train_loops_gpu.py tests/test_full_size_properties_gpu.py -x -q 2>&1 | tail -3
python bench.py --workload wav2vec2 --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
