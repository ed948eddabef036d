"""cProfile of the host side of the Whisper step (where do the ~8 ms/step of Python go?)."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd
from tethys_speech_amd import dist as D, optim, train, whisper
from tethys_speech_amd.data import create_dummy_dataset
dev = "cuda:0"
strategy = D.DataParallelStrategy(0, 1)
model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
opt = optim.Adam(1e-4)
it = iter(create_dummy_dataset(8, device=dev, rank=0, world=1, seed=1234, drop_remainder=True))
for _ in range(3):
    train.distributed_train_step(strategy, model, next(it), opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    train.distributed_train_step(strategy, model, next(it), opt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(18)
