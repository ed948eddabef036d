# HBM traffic of the GEMM kernels: two rocprofv3 passes, one counter each (never combined with other trace domains)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc
rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  TMI_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > $O/$c.log 2>&1
  echo "$c pass done"
done
f=$(find $O/FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find $O/WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $f $w 4 $O/gemm_pmc_traffic.json
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
cat $O/gemm_pmc_traffic.json | tail -5
