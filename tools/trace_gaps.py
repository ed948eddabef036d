"""GPU busy fraction and idle gaps from a rocprofv3 kernel_trace.csv: union of kernel intervals over the
span of the last `steps` steps (steps delimited by the Adam kernel).  usage: trace_gaps.py <dir> [steps]"""
import csv, glob, sys
d = sys.argv[1]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(tr))))
# a step ends with its LAST optimizer launch (the update may be several launches: dense ranges, the row-sparse
# embedding table, or the segmented Adam-with-clipping): consecutive Adam launches with nothing between them are one
# (round 3: early Adam slices run inside backward, so optimizer launches no longer delimit steps: a step STARTS with the
# input transpose of Whisper (feat_cl_kernel) or the FIR filter bank of Wav2Vec2; the last step ends with its last Adam)
starts = [i for i, r in enumerate(rows) if "feat_cl_kernel" in r[2] or ("fir_gn_partial" in r[2])]
starts = [i for k_, i in enumerate(starts) if k_ == 0 or rows[i][0] - rows[starts[k_ - 1]][0] > 1_000_000]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nsteps = min(nsteps, len(starts) - 1)
lo, hi = starts[-nsteps - 1], starts[-1]
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]; gaps = []
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"{nsteps} steps: span {span / 1e6 / nsteps:.3f} ms/step, busy {busy / 1e6 / nsteps:.3f} ms/step ({100.0 * busy / span:.1f} %), "
      f"{len(gaps) / nsteps:.0f} gaps/step totalling {sum(g for g, _ in gaps) / 1e6 / nsteps:.3f} ms/step")
gaps.sort(reverse=True)
for g, n in gaps[:12]:
    print(f"  gap {g / 1e3:7.1f} us before {n[:90]}")
import collections
hist = collections.Counter(min(int(g / 1e3) // 5 * 5, 50) for g, _ in gaps)
print("gap histogram (us bucket: count/step):", {k: round(v / nsteps, 1) for k, v in sorted(hist.items())})
