cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_whisper_step_gpu.py tests/test_workspace_guards_gpu.py -q -x -k "attention or step or workspace" 2>&1 | tail -3
for e in 1 0 1 0; do
  echo "== TMI_ATTN_NO_KSPLIT=$e"
  TMI_ATTN_NO_KSPLIT=$e python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -E "timed" | cut -c1-200
done
