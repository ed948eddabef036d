"""Prints the actual errors behind the tolerance-based step tests (diagnostic: how much margin the bounds have)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper
from tethys_speech_amd.data import create_dummy_dataset
from oracle import whisper_oracle as O, dropout as DO
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_whisper_step_gpu as T

dev = "cuda:0"
for trial in range(3):
    model, ocfg, params = T.build("bf16", T.small_cfg(), dev)
    model.enable_dropout(0.1, 0.1, seed=0xC0FFEE + trial)
    ocfg = O.make_config_like(ocfg, dropout=0.1, attention_dropout=0.1)
    feats, labels = O.create_dummy_pool(seed=13 + trial, n_mels=16, seq_len=47, max_target_length=12, num_samples=3)
    for k in params:
        if k.endswith(".kernel"):
            params[k] = params[k].to(torch.bfloat16).double()
    O.DROPOUT_PROVIDER = DO.HostDropout(0xC0FFEE + trial, 0)
    lr, gr = O.loss_and_grads(params, torch.from_numpy(feats), torch.from_numpy(labels), ocfg)
    O.DROPOUT_PROVIDER = None
    l = float(model.forward_backward(torch.from_numpy(feats).to(dev), torch.from_numpy(labels).to(dev)).item())
    got = model.arena.ref_views(model.arena.g)
    worst = max(float((got[k].double().cpu() - g).norm() / max(float(g.norm()), 1e-2)) for k, g in gr.items())
    print(f"dropout step trial {trial}: |dloss| {abs(l - float(lr)):.2e} (bound 2e-2), worst grad rel {worst:.3e} (bound 6e-2)")

model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
feats, labels = next(iter(create_dummy_dataset(8, device=dev, seed=1234, drop_remainder=True)))
for trial in range(3):
    l8 = float(model.forward_backward(feats, labels).item()); g8 = model.arena.g.clone()
    la = float(model.forward_backward(feats[:4].contiguous(), labels[:4].contiguous()).item()); ga = model.arena.g.clone()
    lb = float(model.forward_backward(feats[4:].contiguous(), labels[4:].contiguous()).item()); gb = model.arena.g.clone()
    print(f"full size trial {trial}: loss {l8:.4f}, |mean halves - l8|/l8 {abs(0.5*(la+lb)-l8)/l8:.2e} (bound 2e-3), grad rel {float((0.5*(ga+gb)-g8).norm()/g8.norm()):.3e} (bound 2e-2)")
