"""Memory traffic of the tmi_gemm kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one
counter per pass), corrected as guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB)
is doubled (128-byte requests tallied at 64 B), WRITE_SIZE (KiB) is taken as is.  Both are the L2's
fabric-side request counters: Infinity-Cache hits are counted, so the figure bounds HBM bytes from above.
usage: pmc_traffic.py <fetch_csv> <write_csv> <steps_profiled> <out_json>"""
import csv, json, sys
from collections import defaultdict

fcsv, wcsv, steps, out = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
GEMM = ("gemm_fast_kernel", "gemm_p8_kernel", "gemm_kernel", "splitk_reduce_kernel")


def load(path, name):
    tot = defaultdict(float); n = defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = next((g for g in GEMM if g in r["Kernel_Name"]), None)
        if k is None:
            continue
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n


ft, fn = load(fcsv, "FETCH_SIZE")
wt, wn = load(wcsv, "WRITE_SIZE")
kib = 1024.0
per = {}
for k in GEMM:
    if fn.get(k, 0) == 0:
        continue
    rd = 2.0 * ft[k] * kib / steps
    wr = wt.get(k, 0.0) * kib / steps
    per[k] = {"launches_per_step": fn[k] / steps, "read_bytes_per_step": rd, "write_bytes_per_step": wr}
launches = sum(v["launches_per_step"] for k, v in per.items() if k != "splitk_reduce_kernel")
total = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in per.values())
import os
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, TMI_WGRAD_STREAM=0) over bench.py " +
                 os.environ.get("PMC_BENCH_ARGS", "--steps 2 --warmup 2") + "; FETCH_SIZE doubled (gfx950 correction), KiB units",
       "steps_profiled": steps, "per_kernel": per, "gemm_launches_per_step": launches,
       "fabric_bytes_per_step": total, "fabric_bytes_per_launch": total / max(1.0, launches),
       "note": "L2 fabric-side bytes (Infinity-Cache hits included): an upper bound on HBM bytes"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
