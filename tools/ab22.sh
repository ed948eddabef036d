cd $GRAFT_REPO_ROOT
for shape in "800 51904 768 nn" "12000 3072 768 nn" "12000 768 2304 nt" "12000 768 768 nt"; do
  python tools/gemm_cfg_probe.py $shape 2>&1 | grep cfg
  for c in 4 5 10 14; do TMI_GEMM_CFG=$c python tools/gemm_cfg_probe.py $shape 2>&1 | grep cfg; done
done
