# kernels of one traced step matching a pattern, with grid sizes: list_kernels.sh <pattern> [bench args...]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PAT=$1; shift
D=gpurun_out/now/_lk; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
python3 tools/step_launches.py $D "$PAT" | cut -c1-160
rm -rf $D
