"""What does a gradient exchange running under backward cost the step on ONE GPU?  

No second GPU is available to the build, so RCCL's copy/reduce kernels are stood in for by a kernel with their
memory behaviour: for every bucket the data-parallel strategy would launch (>= 24 MiB of the fp32 gradient arena,
as soon as backward reports it final), a few-workgroup kernel (RCCL runs one workgroup per channel: 16-64) reads
2 x the bucket and writes 1 x (an 8-GPU ring step reads its own chunk and the peer's and writes the sum: over the
2(N-1) steps of a ring all-reduce the local traffic is ~2(N-1)/N x 3 x bucket bytes; ``--passes`` sets how many
such sweeps each bucket gets: 4 ~ the HBM traffic of an 8-rank ring, 1 ~ a mesh reduce-scatter + all-gather).
It runs on its own stream, ordered after the bucket's producers, and the optimizer waits for it: exactly the
dependencies of the real exchange.  Reports ms/step without it, with it, and the stand-in alone.

  python tools/exchange_overlap_probe.py [--passes 4] [--blocks 32] [--wire bf16]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=4)
ap.add_argument("--blocks", type=int, default=32)
ap.add_argument("--wire", choices=["fp32", "bf16"], default="fp32")
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
os.environ["TMI_GRAD_BLOCKS"] = str(a.blocks)

import torch  # noqa: E402
import tethys_speech_amd  # noqa: E402,F401
from tethys_speech_amd import ops, optim, whisper  # noqa: E402
from tethys_speech_amd.data import create_dummy_dataset  # noqa: E402

dev = "cuda:0"
model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
model.enable_dropout(0.1, 0.1, seed=1)
opt = optim.Adam(1e-4)
it = iter(create_dummy_dataset(8, device=dev, seed=1234, drop_remainder=True))
n = model.arena.numel
wdt = torch.bfloat16 if a.wire == "bf16" else torch.float32
peer = torch.zeros(2 * n, dtype=wdt, device=dev)     # "own chunk" + "peer's chunk" of every bucket
out = torch.zeros(n, dtype=torch.float32, device=dev)
xs = torch.cuda.Stream(device=dev)
BUCKET = 24 << 20


class StandIn:
    def __init__(self, on):
        self.on, self.hi, self.lo = on, n, n

    def ready(self, lo, hi):
        self.lo = lo
        if (self.hi - self.lo) * 4 >= BUCKET:
            self.launch()

    def launch(self):
        if self.on and self.hi > self.lo:
            model._join_side()
            xs.wait_stream(torch.cuda.current_stream())
            prev = ops.set_stream(xs.cuda_stream)
            m = self.hi - self.lo
            for _ in range(a.passes):
                ops.grad_unpack(peer[self.lo:], out[self.lo:self.hi], m, nparts=2, part_stride=n)
            ops.set_stream(prev)
        self.hi = self.lo

    def finish(self):
        self.lo = 0
        self.launch()
        torch.cuda.current_stream().wait_stream(xs)


def step(on):
    s = StandIn(on)
    f, l = next(it)
    model.forward_backward(f, l, grad_ready=s.ready)
    s.finish()
    opt.apply_gradients(model)


def timed(on, k):
    for _ in range(3):
        step(on)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step(on)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


base = timed(False, a.steps)
with_x = timed(True, a.steps)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    prev = ops.set_stream(xs.cuda_stream)
    for _ in range(a.passes):
        ops.grad_unpack(peer, out, n, nparts=2, part_stride=n)
    ops.set_stream(prev)
torch.cuda.synchronize()
alone = (time.perf_counter() - t0) / 5 * 1e3
es = 2 if a.wire == "bf16" else 4
gb = a.passes * (2 * es + 4) * n / 1e9
print(f"wire {a.wire}, {a.passes} sweeps/bucket on {a.blocks} workgroups ({gb:.2f} GB of HBM traffic per step): "
      f"step alone {base:.2f} ms, with the stand-in exchange under backward {with_x:.2f} ms (+{with_x - base:.2f}), "
      f"stand-in alone {alone:.2f} ms ({gb / alone * 1e3:.0f} GB/s)")
