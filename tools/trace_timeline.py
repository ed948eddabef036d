"""One step's kernel timeline from a rocprofv3 kernel_trace.csv: for every dispatch its start offset, duration, queue,
workgroup count, and an estimate of how much of the chip it can occupy (min(1, workgroups / 256) - one-workgroup-per-CU
kernels - refined by nothing: a coarse upper bound).  Prints the timeline (optionally) and the integral
"chip-time not covered": sum over time of (1 - covered fraction), where overlapping kernels add their fractions.
usage: trace_timeline.py <dir> [--list] [--step k]"""
import csv, glob, sys
d = sys.argv[1]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
R = list(csv.DictReader(open(tr)))
def gi(r, *names):
    for n in names:
        if n in r and r[n] != "":
            return int(float(r[n]))
    return 0
rows = []
for r in R:
    gx, gy, gz = gi(r, "Grid_Size_X", "Grid_Size"), gi(r, "Grid_Size_Y") or 1, gi(r, "Grid_Size_Z") or 1
    wx, wy, wz = gi(r, "Workgroup_Size_X", "Workgroup_Size") or 1, gi(r, "Workgroup_Size_Y") or 1, gi(r, "Workgroup_Size_Z") or 1
    wgs = max(1, (gx // wx)) * max(1, gy // wy) * max(1, gz // wz)
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), wgs, wx * wy * wz,
                 gi(r, "LDS_Block_Size"), gi(r, "VGPR_Count", "Arch_VGPR_Count")))
rows.sort()
# a step starts with the input transpose (Whisper: feat_cl_kernel) or the FIR filter bank (Wav2Vec2); optimizer launches
# may sit anywhere inside a step (early Adam slices), so they are no delimiter
starts = [i for i, r in enumerate(rows) if "feat_cl_kernel" in r[2] or ("fir_gn_partial" in r[2])]
starts = [i for k_, i in enumerate(starts) if k_ == 0 or rows[i][0] - rows[starts[k_ - 1]][0] > 1_000_000]
k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else 1
lo, hi = starts[-k - 1], starts[-k]
seg = rows[lo:hi]
t0 = seg[0][0]
t1 = max(r[1] for r in seg)
print(f"step span {(t1 - t0) / 1e3:.1f} us, {len(seg)} dispatches")
queues = sorted(set(r[3] for r in seg))
# coverage integral
ev = []
for s, e, n, q, wgs, wsz, lds, vg in seg:
    # workgroups resident per CU: by LDS (160 KiB) and by waves (32 per CU) - coarse
    per_cu = max(1, min(32 // max(1, wsz // 64), (160 * 1024) // lds if lds else 8, 8))
    frac = min(1.0, wgs / (256.0 * per_cu)) if wgs < 256 * per_cu else 1.0
    frac = max(frac, min(1.0, wgs / 256.0) / per_cu)
    ev.append((s, frac)); ev.append((e, -frac))
ev.sort()
cov, last, idle_area, idle_time = 0.0, t0, 0.0, 0.0
for t, df in ev:
    if t > last:
        c = min(1.0, cov)
        idle_area += (1.0 - c) * (t - last)
        if cov <= 1e-9:
            idle_time += t - last
        last = t
    cov += df
print(f"fully idle {idle_time / 1e3:.1f} us; chip-time not covered (coarse) {idle_area / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f}")
if "--list" in sys.argv:
    for s, e, n, q, wgs, wsz, lds, vg in seg:
        short = n.replace("void (anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")[:70]
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} q{queues.index(q)} wg={wgs:6d}x{wsz:4d} lds={lds // 1024:3d}K v={vg:3d} {short}")
