"""Time tmi_attn_fwd / tmi_attn_bwd on the step's three attention shapes (bf16, head_dim 64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tethys_speech_amd import ops

dev = torch.device("cuda:0")
B, H, HD = 8, 12, 64
D = H * HD


DROP = float(os.environ.get("ATTN_DROPOUT", "0"))


def run(name, Tq, Tk, mask, iters=20):
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda T: (torch.randn(B, T, D, device=dev, generator=g) * 1.0).to(torch.bfloat16)
    q, k, v, do = mk(Tq), mk(Tk), mk(Tk), mk(Tq)
    o = torch.empty_like(q); dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    stats = torch.empty(B, H, Tq, 2, device=dev); delta = torch.empty(B, H, Tq, device=dev)
    sc = HD ** -0.5
    dm = ops.attn_dropmask(dev, B, H, Tq, Tk) if DROP > 0 else None
    Q = (q, 0, Tq * D, D); K = (k, 0, Tk * D, D); V = (v, 0, Tk * D, D); O = (o, 0, Tq * D, D)
    fwd = lambda: ops.attn_fwd(Q, K, V, O, stats, B, H, Tq, Tk, mask, score_scale=sc, dropout_p=DROP, dropout_seed=77, drop_mask=dm)
    bwd = lambda: ops.attn_bwd(Q, K, V, O, stats, (do, 0, Tq * D, D), (dq, 0, Tq * D, D), (dk, 0, Tk * D, D),
                               (dv, 0, Tk * D, D), delta, B, H, Tq, Tk, mask, score_scale=sc, dropout_p=DROP, dropout_seed=77, drop_mask=dm)
    for fn, label, nprod in ((fwd, "fwd", 2), (bwd, "bwd(dq+dkv)", 7)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        fl = 2.0 * B * H * Tq * Tk * HD * nprod
        print(f"p={DROP} {name:10s} {label:12s} Tq={Tq:5d} Tk={Tk:5d} mask={mask} {us:8.1f} us  {fl / us * 1e-6:7.1f} TF/s")


run("enc-self", 1500, 1500, 0)
run("dec-cross", 100, 1500, 0)
run("dec-self", 100, 100, 1)
run("w2v-self", 99, 99, 0)
