cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4p; rm -rf $O; mkdir -p $O
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/serial.log 2>&1
cp $(find $O/_s -name '*kernel_stats.csv' | head -1) $O/serial_kernel_stats.csv
python3 tools/prof_summary.py $O/_s 7 > $O/serial_summary.txt 2>&1
python3 tools/trace_timeline.py $O/_s --list > $O/serial_timeline.txt 2>&1
rm -rf $O/_s
head -50 $O/serial_summary.txt
