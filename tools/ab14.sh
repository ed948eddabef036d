cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/ab14_tests.log 2>&1; tail -3 gpurun_out/ab14_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in whisper wav2vec2; do
  python bench.py --workload $w --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline --dropout off 2>&1 | grep -E "timed" | cut -c1-200
