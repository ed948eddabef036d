cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -3
for c in 10 14; do TMI_GEMM_CFG=$c python tools/gemm_rule_probe.py 768 3072 2>&1 | grep cfg; done
python tools/gemm_rule_probe.py 768 3072 2>&1 | grep cfg
for e in 1 0; do
  echo "== TMI_GEMM_NO_P8_192=$e"
  TMI_GEMM_NO_P8_192=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
