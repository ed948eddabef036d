# re-measure the Wav2Vec2 lines of the round-2 set (after the LayerNorm-backward emission went into its backward)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
prof() {
  tag=$1; shift
  D=$O/_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/$tag.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${tag}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${tag}_gaps.txt 2>&1 || true
  cp $(find $D -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $D
}
TMI_WGRAD_STREAM=0 prof wav2vec2_serial --workload wav2vec2 --steps 4 --warmup 3
prof wav2vec2_overlap --workload wav2vec2 --steps 4 --warmup 3
python3 bench.py --workload wav2vec2 > $O/bench_wav2vec2_n1.json 2> $O/bench_wav2vec2_n1.log
grep timed $O/bench_wav2vec2_n1.log
