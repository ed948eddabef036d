"""Host enqueue time vs device time of one Wav2Vec2-base step (B = 8, 2 s clips)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import dist as D, optim, train, wav2vec2
from tethys_speech_amd.data import W2VDummyDataset
dev = "cuda:0"
strategy = D.DataParallelStrategy(0, 1)
model = wav2vec2.create_full_model("pretraining", "base", device=dev, precision="bf16", seed=1234)
c = model.config
model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=1, act_p=c.activation_dropout)
opt = optim.Adam(3e-5, epsilon=1e-8)
it = iter(W2VDummyDataset(8, device=dev, seed=1234))
rng = np.random.default_rng(1)
negs = [torch.from_numpy(wav2vec2.sample_negative_indices(rng, 8, 100, 100)).to(dev) for _ in range(8)]
for i in range(5):
    train.wav2vec2_train_step(strategy, model, next(it), negs[i % 8], opt)
torch.cuda.synchronize()
host, total = [], []
for i in range(20):
    b = next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train.wav2vec2_train_step(strategy, model, b, negs[i % 8], opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"wav2vec2-base step from an idle device: host enqueue median {host[10]:.2f} ms, enqueue + drain median {total[10]:.2f} ms")
