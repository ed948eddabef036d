cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/kg
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/kg/tests.log 2>&1 || { tail -20 gpurun_out/kg/tests.log; exit 1; }
tail -2 gpurun_out/kg/tests.log
for i in 1 2; do
  TMI_GEMM_NO_KGROUPS=1 python bench.py --workload wav2vec2 --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/kg/w2v_off_$i.json
  python bench.py --workload wav2vec2 --steps 200 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/kg/w2v_on_$i.json
  TMI_GEMM_NO_KGROUPS=1 python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/kg/wh_off_$i.json
  python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/kg/wh_on_$i.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/kg/*.json")):
    try:
        d=json.loads(open(f).read()); print(f, d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
P
