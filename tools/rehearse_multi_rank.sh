# usage: rehearse_multi_rank.sh [N] [steps] [warmup] ["extra bench args", e.g. "--workload wav2vec2"]  (steps 4 warmup 3: the timed steps are launch-plan replays with the collectives as callback nodes)
# bench.py --gpus N (default 4) rehearsed on ONE GPU: the ranks share cuda:0, gloo carries the buckets (RCCL cannot put two ranks on a device).
# Exercises the launch contract, the weak-scaling data pool, the overlapped bucket exchange, MAX-over-ranks timing and the JSON line;
# the rate it prints means nothing (host-staged all-reduce of 591 MB per step, four processes time-slicing one device).
cd $GRAFT_REPO_ROOT
export TETHYS_ONE_DEVICE=1 TETHYS_DIST_BACKEND=gloo
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${1:-4} --master-addr 127.0.0.1 --master-port 29577 \
  bench.py --gpus ${1:-4} --steps ${2:-2} --warmup ${3:-1} --no-cpu-baseline ${4:-} > gpurun_out/rehearse_multi_rank.json 2> gpurun_out/rehearse_multi_rank.log
echo "rc=$?"
cat gpurun_out/rehearse_multi_rank.json | cut -c1-700
grep -E "timed|Error|error" gpurun_out/rehearse_multi_rank.log | head -5
