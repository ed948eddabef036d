# BASELINE configs[4]: Whisper "large" (1280/20h/5120, 32+32 layers, W:880-886), per-GPU batch 8, bf16, one MI355X:
# the bench line (roofline classes included), the serial kernel stats and the MFMA-busy counters.  -> gpurun_out/r5L/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5L; rm -rf $O; mkdir -p $O
python3 bench.py --model_type large --steps 10 --warmup 5 --no-cpu-baseline > $O/bench_whisper_large_n1.json 2> $O/bench_whisper_large_n1.log || { tail -5 $O/bench_whisper_large_n1.log; exit 1; }
echo "bench done"
D=$O/_s
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --model_type large --steps 2 --warmup 4 --no-cpu-baseline --no-roofline > $O/serial.log 2>&1
python3 tools/prof_summary.py $D 6 > $O/whisper_large_serial_summary.txt 2>&1 || true
cp $(find $D -name '*kernel_stats.csv' | head -1) $O/whisper_large_serial_kernel_stats.csv
rm -rf $D
echo "serial trace done"
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/_a -- python3 bench.py --model_type large --steps 1 --warmup 3 --no-cpu-baseline --no-roofline --no-plan > $O/a.log 2>&1
python3 tools/pmc_kernel_counters.py $(find $O/_a -name '*counter_collection.csv' | head -1) > $O/whisper_large_pmc_mfma.txt 2>&1
rm -rf $O/_a
head -3 $O/bench_whisper_large_n1.log; tail -c 1500 $O/bench_whisper_large_n1.json; head -12 $O/whisper_large_serial_summary.txt; head -14 $O/whisper_large_pmc_mfma.txt
