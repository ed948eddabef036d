# usage: trace_one.sh <kernel-substring> [env...]: average duration of matching kernels in a short traced bench run
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=$1; shift
D=gpurun_out/now/_one
rm -rf $D
env "$@" true
( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
f=$(find $D -name '*kernel_stats.csv' | head -1)
grep -i "$PAT" $f | cut -c1-200
rm -rf $D
