"""tmi_xent_fwd_bwd on the step's logits ([800, 51904] bf16, V = 51865): time per launch and bytes moved (one read + one write)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops
dev = "cuda:0"
B, S, V, ld = 8, 100, 51865, 51904
torch.manual_seed(0)
src = (torch.randn(B * S, ld, device=dev) * 2).to(torch.bfloat16)
labels = torch.randint(0, V, (B, S), device=dev, dtype=torch.int32)
row_loss = torch.empty(B * S, device=dev)
bufs = [src.clone() for _ in range(8)]   # (the kernel overwrites its input: rotate copies so every launch reads logits-like data)
f = lambda i: ops.xent_fwd_bwd(bufs[i % 8], ld, labels, row_loss, B, S, V, 1.0 / (B * (S - 1)))
for i in range(8):
    f(i)
torch.cuda.synchronize()
for i in range(8):
    bufs[i].copy_(src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(8):
    f(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 8
print(f"xent_fwd_bwd [800, {ld}] bf16: {us:.1f} us  {2 * B * S * ld * 2 / us / 1e6:.2f} TB/s (read + write)  loss {float(row_loss.sum() / (B * (S - 1))):.4f}")
