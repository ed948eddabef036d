# which kernels does the vendor library run for the step's GEMM shapes?  (rocprofv3 kernel trace of tools/lib_gemm_probe.py)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=gpurun_out/_libk
rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/lib_gemm_probe.py > gpurun_out/lib_kernel_probe.log 2>&1
python3 - <<'P' > gpurun_out/lib_kernel_names.txt
import csv, glob, collections
f = glob.glob("gpurun_out/_libk/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.startswith("Cijk") or "gemm" in n.lower():
        k = (n, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (n, g, w, lds), (c, t) in agg.items():
    print(f"n={c:4d} avg={t / c:8.1f} us grid={g} wg={w} lds={lds}  {n[:260]}")
P
rm -rf $D
cat gpurun_out/lib_kernel_names.txt
