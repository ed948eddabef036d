cd $GRAFT_REPO_ROOT
for s in "X=0" "TMI_GEMM_P8_PERSIST=0" "TMI_GEMM_LEAN_EPI=0" "TMI_GEMM_NO_P8=1" "TMI_ATTN_NO_KSPLIT=1" "TMI_GEMM_NO_KGROUPS=1" "TMI_GEMM_GENERIC=1"; do
  echo "== $s"; env $s python tools/fwd_determinism.py 2>&1 | grep "bf16 model 0"
done
