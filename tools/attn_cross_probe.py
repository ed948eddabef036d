"""Where the decoder's cross-attention backward spends its time: tmi_attn_bwd on (B, Tq) variations of the [*, 12, Tq, 1500]
shape, run under `rocprofv3 --kernel-trace` (tools/attn_cross_probe.sh prints per-kernel averages per configuration).
Fixed cost per workgroup vs cost per query tile vs dependence on the number of workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")
H, HD, Tk = 12, 64, 1500
D = H * HD
DROP = float(os.environ.get("ATTN_DROPOUT", "0"))
CONFIGS = [(8, 100), (8, 64), (8, 128), (8, 256), (8, 512), (4, 100), (2, 100), (1, 100)]


def run(B, Tq, iters=10):
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda T: torch.randn(B, T, D, device=dev, generator=g).to(torch.bfloat16)
    q, k, v, do = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
    o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    stats = torch.empty(B, H, Tq, 2, device=dev); delta = torch.empty(B, H, Tq, device=dev)
    sc = HD ** -0.5
    dm = ops.attn_dropmask(dev, B, H, Tq, Tk) if DROP > 0 else None
    Q = (q, 0, Tq * D, D); K = (k, 0, Tk * D, D); V = (v, 0, Tk * D, D); O = (o, 0, Tq * D, D)
    ops.attn_fwd(Q, K, V, O, stats, B, H, Tq, Tk, 0, score_scale=sc, dropout_p=DROP, dropout_seed=77, drop_mask=dm)
    for _ in range(iters):
        ops.attn_bwd(Q, K, V, O, stats, (do, 0, Tq * D, D), (dq, 0, Tq * D, D), (dk, 0, Tk * D, D), (dv, 0, Tk * D, D), delta,
                     B, H, Tq, Tk, 0, score_scale=sc, dropout_p=DROP, dropout_seed=77, drop_mask=dm)
    torch.cuda.synchronize()


for B, Tq in CONFIGS:
    run(B, Tq)
print("configs", CONFIGS)
