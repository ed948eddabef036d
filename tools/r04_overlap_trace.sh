# One overlapped-step kernel trace (the default step: weight gradients on the side stream) -> per-step timeline with queue ids.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4o; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/_s -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline > $O/overlap.log 2>&1
python3 tools/trace_timeline.py $O/_s --list > $O/overlap_timeline.txt 2>&1
python3 tools/trace_gaps.py $O/_s > $O/overlap_gaps.txt 2>&1
rm -rf $O/_s
head -3 $O/overlap_timeline.txt
