cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm" > gpurun_out/t_ln.log 2>&1 || { tail -30 gpurun_out/t_ln.log; exit 1; }
tail -1 gpurun_out/t_ln.log
for cfg in "TMI_LN_FWD_ROWS=3" "TMI_LN_FWD_ROWS=1 TMI_LN_FWD_BLOCKS=1024" "TMI_LN_FWD_ROWS=2" "TMI_LN_FWD_ROWS=3 TMI_LN_PF=2"; do
  echo "== $cfg"; env $cfg python tools/ln_bench.py 2>/dev/null | grep "12000,  768\|800"
done
