set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_wav2vec2_gpu.py -x -q -k "fir or step or curve or golden" 2>&1 | tail -3
TMI_WGRAD_STREAM=0 bash tools/profile_one.sh w2vser3 7 --workload wav2vec2 --steps 4 --warmup 3 > /dev/null
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_w2vser3_kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if 'fir' in n:
        print(n[:70], r['Calls'], float(r['TotalDurationNs'])/int(r['Calls'])/1e3)
PY
python bench.py --workload wav2vec2 --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
