# wall-time phases of one overlapped training step from a kernel trace -> gpurun_out/now/phases.txt
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/now; mkdir -p $O
D=$O/_ph
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline > $O/ph.log 2>&1
python3 tools/step_launches.py $D "" > $O/overlap_launches.txt
rm -rf $D
python3 - <<'PY'
import re
rows=[]
for l in open('gpurun_out/now/overlap_launches.txt'):
    m=re.match(r"\s*([\d.]+) us\s+\+\s*([\d.]+) us\s+grid\s+(\d+)\s+(.*)",l)
    if m: rows.append((float(m.group(1)),float(m.group(2)),int(m.group(3)),m.group(4)))
def first(pred, start=0):
    for i in range(start,len(rows)):
        if pred(rows[i]): return i
    return None
end=max(t+d for t,d,_,_ in rows)
# markers: first decoder-sized attention fwd (grid 256 threads*... small), xent, first encoder dkv after xent, adam
i_xent=first(lambda r:'xent' in r[3])
i_dec=first(lambda r:'attn_fwd' in r[3] and r[2]<=1024)
i_encb=first(lambda r:'attn_bwd_dq' in r[3] and r[2]>2000, i_xent)
i_adam=first(lambda r:'adam' in r[3])
print(f"step span {end/1e3:.2f} ms: encoder fwd until {rows[i_dec][0]/1e3:.2f}; decoder fwd until {rows[i_xent][0]/1e3:.2f} (xent); "
      f"first encoder attention backward at {rows[i_encb][0]/1e3:.2f}; Adam at {rows[i_adam][0]/1e3:.2f}")
PY
