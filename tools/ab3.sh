set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_whisper_step_gpu.py tests/test_two_rank_gpu.py tests/test_full_size_properties_gpu.py -x -q 2>&1 | tail -5
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "== DEC_EARLY=$1 KV_PER_LAYER=$2"
  TMI_DEC_EARLY=$1 TMI_KV_PER_LAYER=$2 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
