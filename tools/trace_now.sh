# quick kernel-trace summaries of the current tree -> gpurun_out/now/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/now
rm -rf $O; mkdir -p $O
prof() {
  tag=$1; shift
  D=$O/_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/$tag.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${tag}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${tag}_gaps.txt 2>&1 || true
  rm -rf $D
}
TMI_WGRAD_STREAM=0 prof step_serial --steps 4 --warmup 3
prof step_overlap --steps 4 --warmup 3
TMI_WGRAD_STREAM=0 prof wav2vec2_serial --workload wav2vec2 --steps 4 --warmup 3
prof wav2vec2_overlap --workload wav2vec2 --steps 4 --warmup 3
echo done
