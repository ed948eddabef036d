cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
export ATTN_DROPOUT=0.1
{
echo "== default"; python tools/attn_bench.py 2>&1 | grep enc-self
for f in 2 4 5; do echo "== FWD_OCC=$f"; TMI_ATTN_FWD_OCC=$f python tools/attn_bench.py 2>&1 | grep "enc-self *fwd"; done
for q in 3 4; do echo "== DQ_OCC=$q DKV default"; TMI_ATTN_DQ_OCC=$q python tools/attn_bench.py 2>&1 | grep "enc-self *bwd"; done
for k in 3 4; do echo "== DKV_OCC=$k DQ default"; TMI_ATTN_DKV_OCC=$k python tools/attn_bench.py 2>&1 | grep "enc-self *bwd"; done
echo "== DQ 3 DKV 3"; TMI_ATTN_DQ_OCC=3 TMI_ATTN_DKV_OCC=3 python tools/attn_bench.py 2>&1 | grep "enc-self *bwd"
export ATTN_DROPOUT=0
echo "== no dropout default"; python tools/attn_bench.py 2>&1 | grep enc-self
for f in 4 5; do echo "== nodrop FWD_OCC=$f"; TMI_ATTN_FWD_OCC=$f python tools/attn_bench.py 2>&1 | grep "enc-self *fwd"; done
} > gpurun_out/r3i/occ.txt 2>&1
cat gpurun_out/r3i/occ.txt
