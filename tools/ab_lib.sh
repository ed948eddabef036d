# A/B of two builds of libtethys_mi.so on one box: abx/lib_old.so vs abx/lib_new.so (copied over the in-tree library in turn)
cd $GRAFT_REPO_ROOT
L=tethys-speech_amd/libtethys_mi.so
cp abx/lib_new.so $L
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or flash or dropout or layernorm" > gpurun_out/ab_lib_tests.log 2>&1 || { tail -30 gpurun_out/ab_lib_tests.log; exit 1; }
tail -1 gpurun_out/ab_lib_tests.log
for which in old new old new; do
  cp abx/lib_$which.so $L
  echo "== $which"
  ATTN_DROPOUT=0.1 python tools/attn_bench.py 2>/dev/null | grep "enc-self\|dec-cross"
  python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; print('step', round(json.loads(sys.stdin.read())['ms_per_step'],3))"
done
cp abx/lib_new.so $L
