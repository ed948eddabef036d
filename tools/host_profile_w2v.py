"""cProfile of the host side of the Wav2Vec2-base step (B = 8, 2 s clips): where the ~4 ms of enqueue time go."""
import cProfile, os, pstats, sys
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
pr = cProfile.Profile()
pr.enable()
for i in range(50):
    train.wav2vec2_train_step(strategy, model, next(it), negs[i % 8], opt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
