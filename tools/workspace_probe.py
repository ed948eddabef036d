"""Diagnostics: run Wav2Vec2 / Whisper steps with every torch.empty workspace buffer filled with NaN (TMI_WS_POISON=1 or a
comma list of buffer names): a buffer read before it is written turns the loss or the gradients NaN."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_wav2vec2_gpu as TW
import tethys_speech_amd
from tethys_speech_amd import optim
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model, ocfg, params = TW.build(prec, dev)
pool = TW.V.create_dummy_pool(seed=3, num_samples=5, length=400)
T = TW.V.feature_lengths(ocfg, 400)[-1]
rng = np.random.default_rng(42)
model.neg_per_time = True
for s, rows in enumerate([pool[0:2], pool[4:5], pool[2:4]]):
    neg = TW.V.sample_negative_indices_roll(rng, T, ocfg.num_negatives)
    loss = model.forward_backward(torch.from_numpy(np.ascontiguousarray(rows)).to(dev), torch.from_numpy(neg).to(dev), num_replicas=1)
    torch.cuda.synchronize()
    g = model.arena.ref_views(model.arena.g)
    bad = [k for k, v in g.items() if not torch.isfinite(v).all()]
    print("step", s, "B", rows.shape[0], "loss", float(loss.item()), "non-finite grads:", bad[:6], len(bad))
    model.arena.g.zero_()
    if os.environ.get("TMI_WS_GUARD"): print("  overrun buffers:", model.check_workspace_guards())
