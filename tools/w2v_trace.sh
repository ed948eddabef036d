cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/w2vtr; rm -rf $O; mkdir -p $O
prof() {
  tag=$1; shift
  D=$O/_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/$tag.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${tag}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${tag}_gaps.txt 2>&1 || true
  python3 tools/trace_timeline.py $D --list > $O/${tag}_timeline.txt 2>&1 || true
  cp $(find $D -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $D
}
prof wav2vec2_overlap --workload wav2vec2 --steps 5 --warmup 4
TMI_WGRAD_STREAM=0 prof wav2vec2_serial --workload wav2vec2 --steps 5 --warmup 4
prof wav2vec2_overlap_noplan --workload wav2vec2 --steps 5 --warmup 4 --no-plan
head -16 $O/wav2vec2_overlap_gaps.txt; head -3 $O/wav2vec2_overlap_noplan_gaps.txt; head -3 $O/wav2vec2_serial_gaps.txt
