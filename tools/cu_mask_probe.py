"""What a hipExtStreamCreateWithCUMask stream costs on this part: one chip-filling GEMM (12000 x 3072 x 768) and one HBM-bound
elementwise pass timed on the default stream, on a stream with every CU enabled, and with every p-th CU disabled."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
ncu = torch.cuda.get_device_properties(dev).multi_processor_count


def masked_stream(keep):
    words = (ncu + 31) // 32
    bits = [0] * words
    for i in range(ncu):
        if keep(i):
            bits[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), (ctypes.c_uint32 * words)(*bits))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


bf = torch.bfloat16
M, N, K = 12000, 3072, 768
A = torch.randn(M, K, device=dev).to(bf); B = torch.randn(K, N, device=dev).to(bf); C = torch.empty(M, N, device=dev, dtype=bf)
x = torch.randn(64 << 20, device=dev)
streams = {"default": torch.cuda.current_stream(), "plain side stream": torch.cuda.Stream(device=dev), "masked, all CUs": masked_stream(lambda i: True),
           "masked, i%5!=4": masked_stream(lambda i: i % 5 != 4), "masked, i%2==0": masked_stream(lambda i: i % 2 == 0),
           "masked, i<128": masked_stream(lambda i: i < 128), "masked, i<32": masked_stream(lambda i: i < 32)}
for name, st in streams.items():
    prev = ops.set_stream(st.cuda_stream)
    with torch.cuda.stream(st):
        for what, fn in (("gemm", lambda: ops.gemm(A, B, C, M, N, K, K, 1, N, 1, N)), ("x*2 (512 MB)", lambda: x.mul_(1.0))):
            for _ in range(3):
                fn()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                fn()
            e1.record(st)
            st.synchronize()
            print(f"{name:22s} {what:14s} {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
    ops.set_stream(prev)
