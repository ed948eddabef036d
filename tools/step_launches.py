"""Per-launch durations of one training step from a rocprofv3 kernel_trace.csv (launch order), filtered by a kernel-name
substring.  usage: step_launches.py <dir> <substring> [step_from_end]"""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")),
                r.get("Workgroup_Size_X", "")) for r in csv.DictReader(open(tr))))
adam_all = [i for i, r in enumerate(rows) if "adam_" in r[2] and "kernel" in r[2]]
adam = [i for k, i in enumerate(adam_all) if k + 1 == len(adam_all) or adam_all[k + 1] != i + 1]
lo, hi = adam[-2] + 1, adam[-1] + 1
t0 = rows[lo][0]
for s, e, n, gx, wx in rows[lo:hi]:
    if pat in n:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  grid {gx:>8}  {n[:70]}")
