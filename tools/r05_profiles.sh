# Round-5 measurement set (one MI355X): rocprofv3 summaries, PMC traffic, timelines, bench lines -> gpurun_out/r05/ (the files
# cited by DESIGN.md are copied to profiles/ by hand).  Run on the GPU box from the repo root: bash tools/r05_profiles.sh
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05
rm -rf $O; mkdir -p $O
prof() {  # tag [env...] -- bench args
  tag=$1; shift
  D=$O/_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > $O/$tag.log 2>&1
  python3 tools/prof_summary.py $D 7 > $O/${tag}_summary.txt 2>&1 || true
  python3 tools/trace_gaps.py $D > $O/${tag}_gaps.txt 2>&1 || true
  python3 tools/trace_timeline.py $D --list > $O/${tag}_timeline.txt 2>&1 || true
  cp $(find $D -name '*kernel_stats.csv' | head -1) $O/${tag}_kernel_stats.csv
  rm -rf $D
}
TMI_WGRAD_STREAM=0 prof step_serial --steps 4 --warmup 3
echo "serial done"
prof step_overlap --steps 4 --warmup 3
TMI_WGRAD_STREAM=0 prof wav2vec2_serial --workload wav2vec2 --steps 4 --warmup 3
prof wav2vec2_overlap --workload wav2vec2 --steps 4 --warmup 3
TMI_WGRAD_STREAM=0 prof fp32_serial --precision fp32 --steps 3 --warmup 2
echo "traces done"
pmc() {  # tag -- bench args
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/_pmc_$c -- python3 bench.py "$@" --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_${tag}_$c.log 2>&1
  done
  f=$(find $O/_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
  w=$(find $O/_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  PMC_BENCH_ARGS="$* --steps 2 --warmup 2" python3 tools/pmc_traffic.py $f $w 4 $O/r05_${tag}_gemm_pmc_traffic.json > /dev/null
  rm -rf $O/_pmc_FETCH_SIZE $O/_pmc_WRITE_SIZE
}
pmc whisper
pmc wav2vec2_base --workload wav2vec2
echo "pmc done"
python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.log
echo "bench default done"
python3 bench.py --dropout off --no-cpu-baseline > $O/bench_n1_dropout_off.json 2> $O/bench_n1_dropout_off.log
python3 bench.py --precision fp32 --steps 40 --no-cpu-baseline > $O/bench_n1_fp32.json 2> $O/bench_n1_fp32.log
python3 bench.py --workload wav2vec2 > $O/bench_wav2vec2_n1.json 2> $O/bench_wav2vec2_n1.log
python3 bench.py --workload whisper_single --batch_size 4 --steps 100 > $O/bench_whisper_single_n1.json 2> $O/bench_whisper_single_n1.log || true
echo "bench lines done"
python3 tools/host_step_time.py > $O/host_step_time.txt 2>&1 || true
python3 tools/host_step_time.py wav2vec2 >> $O/host_step_time.txt 2>&1 || true
python3 tools/lib_gemm_probe.py > $O/lib_gemm_probe.txt 2>&1 || true
python3 tools/gemm_epi_probe.py > $O/gemm_epi_probe.txt 2>&1 || true
python3 tools/gemm_f32_probe.py > $O/gemm_f32_probe.txt 2>&1 || true
python3 tools/attn_bench.py > $O/attn_bench.txt 2>&1 || true
ATTN_DROPOUT=0.1 python3 tools/attn_bench.py >> $O/attn_bench.txt 2>&1 || true
python3 tools/ln_bench.py > $O/ln_bench.txt 2>&1 || true
# (tools/p8_stamps.py needs a library built with `make EXPERIMENTS=1`: the stamps are compiled out of the shipped one)
echo "all done"
ls $O
