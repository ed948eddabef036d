set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/now
mkdir -p $O
D=$O/_t
TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/t.log 2>&1
python3 tools/step_launches.py $D "" > $O/all_launches.txt
rm -rf $D
