set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-trace}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline ${@:2} > $O/run.log 2>&1
python3 tools/trace_timeline.py $O/raw --list > $O/timeline.txt 2>&1 || true
python3 tools/trace_gaps.py $O/raw > $O/gaps.txt 2>&1 || true
python3 tools/prof_summary.py $O/raw 4 > $O/summary.txt 2>&1 || true
rm -rf $O/raw
head -3 $O/timeline.txt; head -3 $O/gaps.txt
