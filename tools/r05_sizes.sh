# bench.py over the reference's other size-table rows (W:859-886), per-GPU batch 8, dropout on / off -> gpurun_out/r05_bench_sizes.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bench_sizes.jsonl; rm -f $O
for mt in tiny base medium; do
  for dr in reference off; do
    python3 bench.py --model_type $mt --dropout $dr --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 >> $O
    tail -1 $O | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mt', '$dr', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'audio-s/s', round(d['config']['step_tflops'] or 0), 'TF/s')"
  done
done
