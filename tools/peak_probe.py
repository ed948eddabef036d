"""Measured ceilings of the box, printed next to the vendor figures the rooflines are priced against
("peaks to normalise against must be measured on the box").

  * bf16 GEMM: the vendor library (torch.matmul -> hipBLASLt/rocBLAS) and tmi_gemm on square problems;
  * HBM: device-to-device copy and a stream triad a = b + s*c (library element-wise kernels), 1 GiB arrays.

usage: python tools/peak_probe.py [out.json]
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops

dev = "cuda:0"
VENDOR = {"bf16_dense_tflops": 2500.0, "hbm_gbs": 8000.0}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def main():
    out = {"device": torch.cuda.get_device_name(0), "vendor": VENDOR, "gemm_bf16": [], "hbm": []}
    for n in (4096, 8192, 16384):
        a = torch.randn(n, n, device=dev).to(torch.bfloat16)
        b = torch.randn(n, n, device=dev).to(torch.bfloat16)
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        fl = 2.0 * n ** 3
        t_lib = timed(lambda: torch.matmul(a, b, out=c), 10)
        t_tmi = timed(lambda: ops.gemm(a, b, c, n, n, n, n, 1, n, 1, n), 10)
        row = {"n": n, "library_tflops": fl / t_lib / 1e12, "tmi_gemm_tflops": fl / t_tmi / 1e12}
        out["gemm_bf16"].append(row)
        print(f"bf16 GEMM {n}^3: library {row['library_tflops']:7.1f} TF/s, tmi_gemm {row['tmi_gemm_tflops']:7.1f} TF/s "
              f"(vendor dense peak {VENDOR['bf16_dense_tflops']:.0f})", flush=True)
        del a, b, c
    n = 1 << 28  # 1 GiB of fp32
    x = torch.randn(n, device=dev)
    y = torch.randn(n, device=dev)
    z = torch.empty(n, device=dev)
    t = timed(lambda: z.copy_(x), 10)
    out["hbm"].append({"kernel": "copy 1 GiB (read + write)", "gbs": 2 * 4 * n / t / 1e9})
    t = timed(lambda: torch.add(x, y, alpha=0.5, out=z), 10)
    out["hbm"].append({"kernel": "triad a = b + s*c (2 reads + 1 write)", "gbs": 3 * 4 * n / t / 1e9})
    t = timed(lambda: z.zero_(), 10)
    out["hbm"].append({"kernel": "fill 1 GiB (write only)", "gbs": 4 * n / t / 1e9})
    for r in out["hbm"]:
        print(f"HBM {r['kernel']:40s} {r['gbs']:8.1f} GB/s (vendor {VENDOR['hbm_gbs']:.0f})", flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
