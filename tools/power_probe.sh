# GPU power / clocks while the bench step runs (rocm-smi sampled every ~0.5 s beside bench.py) -> gpurun_out/power.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/power.txt
( rocm-smi --showpower --showclocks --showtemp 2>&1 | head -40 ) > $O
echo "=== under load" >> $O
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-roofline "$@" > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | tr '\n' ';' >> $O
  echo >> $O
  sleep 1
done
wait $BP
grep timed gpurun_out/power_bench.log >> $O
