"""Where the host's ~7 ms per Whisper step go: cProfile over 30 pipelined steps (device kept busy, as in the bench)."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import dist as D, optim, train, whisper
from tethys_speech_amd.data import create_dummy_dataset
dev = "cuda:0"
strategy = D.DataParallelStrategy(0, 1)
model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
model.enable_dropout(0.1, 0.1, seed=1)
opt = optim.Adam(1e-4)
it = iter(create_dummy_dataset(8, device=dev, rank=0, world=1, seed=1234, drop_remainder=True))
for _ in range(5):
    train.distributed_train_step(strategy, model, next(it), opt, pipelined=True)
torch.cuda.synchronize()
N = 30
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    train.distributed_train_step(strategy, model, next(it), opt, pipelined=True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(28)
print(s.getvalue().replace("/root/repo/", ""))
