# usage: bash tools/profile_one.sh <tag> <steps-total> <bench args...>   -> gpurun_out/prof_<tag>_{summary,gaps}.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; total=$2; shift 2
O=gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py "$@" --no-cpu-baseline --no-roofline > gpurun_out/prof_$tag.log 2>&1
python3 tools/prof_summary.py $O $total > gpurun_out/prof_${tag}_summary.txt 2>&1 || true
python3 tools/trace_gaps.py $O > gpurun_out/prof_${tag}_gaps.txt 2>&1 || true
cp $(find $O -name '*kernel_stats.csv' | head -1) gpurun_out/prof_${tag}_kernel_stats.csv
rm -rf $O
head -30 gpurun_out/prof_${tag}_summary.txt | cut -c1-160
