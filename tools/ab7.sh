set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TMI_WGRAD_STREAM=0 bash tools/profile_one.sh w2vser2 7 --workload wav2vec2 --steps 4 --warmup 3 > /dev/null
grep -E "fir_|gn_|total kernel|adam|segment|colsum|ln_|gemm_kernel" gpurun_out/prof_w2vser2_summary.txt | cut -c1-150
