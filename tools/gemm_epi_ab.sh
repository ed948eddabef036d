# tools/gemm_epi_probe.py once per library setting (the switches are read once per process), two rounds.
# Usage on the GPU box: bash tools/gemm_epi_ab.sh "SETTING A" "SETTING B" ...   ("default" = no variables)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for setting in "$@"; do
    if [ "$setting" = default ]; then PROBE_TAG="default" python tools/gemm_epi_probe.py 2>/dev/null
    else env $setting PROBE_TAG="$setting" python tools/gemm_epi_probe.py 2>/dev/null; fi
  done
done
