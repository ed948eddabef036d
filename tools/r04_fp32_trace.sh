# fp32 parity path: bench line + rocprofv3 kernel stats (serial) -> gpurun_out/r4fp32/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4fp32; rm -rf $O; mkdir -p $O
python3 bench.py --precision fp32 --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_fp32.json 2> $O/bench_fp32.log
cat $O/bench_fp32.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 ms/step', d['ms_per_step'])"
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s -- python3 bench.py --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/serial.log 2>&1
cp $(find $O/_s -name '*kernel_stats.csv' | head -1) $O/fp32_serial_kernel_stats.csv
python3 tools/prof_summary.py $O/_s 5 > $O/fp32_serial_summary.txt 2>&1
rm -rf $O/_s
head -24 $O/fp32_serial_summary.txt
