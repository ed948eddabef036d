set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof5
rm -rf $O; mkdir -p $O
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/serial.log 2>&1
echo serial done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/overlap -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/overlap.log 2>&1
echo overlap done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/w2v -- python3 bench.py --workload wav2vec2 --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/w2v.log 2>&1
echo w2v done
for n in serial overlap w2v; do
  f=$(find $O/$n -name '*kernel_stats.csv' | head -1)
  cp $f $O/${n}_kernel_stats.csv
  python3 tools/prof_summary.py $O/$n 7 > $O/${n}_summary.txt 2>&1 || true
  t=$(find $O/$n -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_gaps.py $O/$n > $O/${n}_gaps.txt 2>&1 || true
  rm -rf $O/$n
done
ls -la $O
