# (the occupancy switches TMI_ATTN_FWD_OCC / _DQ_OCC / _DKV_OCC exist in `make EXPERIMENTS=1` builds only; the shipped library ignores them)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "flash" > gpurun_out/t_flash.log 2>&1 || { tail -30 gpurun_out/t_flash.log; exit 1; }
tail -1 gpurun_out/t_flash.log
for cfg in "TMI_ATTN_XCD=1" "TMI_ATTN_XCD=0" "TMI_ATTN_XCD=1 TMI_ATTN_FWD_OCC=2" "TMI_ATTN_XCD=1 TMI_ATTN_FWD_OCC=4" "TMI_ATTN_XCD=1 TMI_ATTN_DQ_OCC=3 TMI_ATTN_DKV_OCC=3" "TMI_ATTN_XCD=1"; do
  echo "== $cfg"
  env $cfg ATTN_DROPOUT=0.1 python tools/attn_bench.py 2>/dev/null | grep "enc-self\|dec-cross"
  env $cfg python tools/attn_bench.py 2>/dev/null | grep "enc-self"
done
