# kernel trace of the Wav2Vec2-base step (serial: every kernel alone; overlap: as timed) -> gpurun_out/$1/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-w2vtrace}
mkdir -p $O
for mode in serial overlap; do
  D=$O/_$mode
  if [ $mode = serial ]; then export TMI_WGRAD_STREAM=0; else unset TMI_WGRAD_STREAM; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload wav2vec2 --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/$mode.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${mode}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${mode}_gaps.txt 2>&1 || true
  python3 tools/trace_timeline.py $D --list > $O/${mode}_timeline.txt 2>&1 || true
  rm -rf $D
done
