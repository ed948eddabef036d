# bench.py over the reference's other size-table rows (per-GPU batch 8) -> gpurun_out/r03_bench_sizes.jsonl
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_bench_sizes.jsonl
: > $O
for m in tiny base medium large; do
  for d in reference off; do
    python bench.py --model_type $m --dropout $d --steps 40 --warmup 3 --no-cpu-baseline 2>> gpurun_out/r03_sizes.log | tail -1 >> $O
    tail -1 $O | cut -c1-200
  done
done
