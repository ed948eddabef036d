set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_wav2vec2_gpu.py -x -q -k "adam or curve" 2>&1 | tail -2
python bench.py --workload wav2vec2 --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
TMI_WGRAD_STREAM=0 bash tools/profile_one.sh w2vser4 7 --workload wav2vec2 --steps 4 --warmup 3 > /dev/null
grep -E "adam|segment|total" gpurun_out/prof_w2vser4_summary.txt | cut -c1-150
