# kernel trace of the fp32 (parity mode) Whisper step -> gpurun_out/fp32trace/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/fp32trace
mkdir -p $O
D=$O/_t
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --precision fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $O/run.log 2>&1
python3 tools/prof_summary.py $D 5 > $O/summary.txt 2>&1 || true
rm -rf $D
head -40 $O/summary.txt
