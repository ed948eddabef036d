# Round-3 measurement set (derived from tools/r02_profiles.sh)
# Round-3 measurement set (one MI355X): rocprofv3 summaries, PMC traffic, bench lines -> gpurun_out/r03/ (copied to profiles/ by hand)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
rm -rf $O; mkdir -p $O
prof() {  # tag steps env... -- bench args
  tag=$1; shift
  D=$O/_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/$tag.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${tag}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${tag}_gaps.txt 2>&1 || true
  cp $(find $D -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $D
}
TMI_WGRAD_STREAM=0 prof step_serial --steps 4 --warmup 3
echo "serial done"
prof step_overlap --steps 4 --warmup 3
TMI_WGRAD_STREAM=0 prof wav2vec2_serial --workload wav2vec2 --steps 4 --warmup 3
prof wav2vec2_overlap --workload wav2vec2 --steps 4 --warmup 3
echo "traces done"
pmc() {  # tag -- bench args
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/_pmc_$c -- python3 bench.py "$@" --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_${tag}_$c.log 2>&1
  done
  f=$(find $O/_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  w=$(find $O/_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  PMC_BENCH_ARGS="$* --steps 2 --warmup 2" python3 tools/pmc_traffic.py $f $w 4 $O/r03_${tag}_gemm_pmc_traffic.json > /dev/null
  rm -rf $O/_pmc_FETCH_SIZE $O/_pmc_WRITE_SIZE
}
pmc whisper
pmc wav2vec2_base --workload wav2vec2
echo "pmc done"
mkdir -p profiles
cp $O/r03_whisper_gemm_pmc_traffic.json $O/r03_wav2vec2_base_gemm_pmc_traffic.json profiles/ 2>/dev/null || true
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.log
echo "bench default done"
python3 bench.py --dropout off --no-cpu-baseline > $O/bench_n1_dropout_off.json 2> $O/bench_n1_dropout_off.log
python3 bench.py --precision fp32 --steps 40 --no-cpu-baseline > $O/bench_n1_fp32.json 2> $O/bench_n1_fp32.log
python3 bench.py --workload wav2vec2 > $O/bench_wav2vec2_n1.json 2> $O/bench_wav2vec2_n1.log
echo "bench lines done"
python3 tools/cli_path_rate.py > $O/cli_path.txt 2>&1 || true
python3 tools/host_step_time.py >> $O/cli_path.txt 2>&1 || true
python3 tools/host_step_time_w2v.py >> $O/cli_path.txt 2>&1 || true
python3 tools/lib_gemm_probe.py > $O/lib_gemm_probe.txt 2>&1 || true
python3 tools/gemm_f32_probe.py > $O/gemm_f32_probe.txt 2>&1 || true
for c in 1 2 4 8; do TMI_GEMM_P8_MAXSPLIT=$c python3 tools/wgrad_split_probe.py 2>/dev/null; done > $O/wgrad_split_probe.txt || true
rocprofv3 --kernel-trace --output-format csv -d $O/_tl -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/timeline.log 2>&1 && python3 tools/trace_timeline.py $O/_tl --list > $O/step_timeline.txt 2>&1 || true
rm -rf $O/_tl
python3 bench.py --workload whisper_single --batch_size 4 --steps 100 > $O/bench_whisper_single_n1.json 2> $O/bench_whisper_single_n1.log || true
echo "all done"
ls $O
