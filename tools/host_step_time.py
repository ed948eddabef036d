"""Host time to ENQUEUE one Whisper small-ref step (device idle at the start of each: the launch queue never
fills, so this is pure host work: Python + ctypes + HIP runtime), beside the device time of the same step."""
import os, sys, time
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
    train.distributed_train_step(strategy, model, next(it), opt)
torch.cuda.synchronize()
host, total = [], []
for _ in range(20):
    b = next(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train.distributed_train_step(strategy, model, b, opt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
host.sort(); total.sort()
print(f"one step from an idle device: host enqueue median {host[10]:.2f} ms (min {host[0]:.2f}), enqueue + drain median {total[10]:.2f} ms (min {total[0]:.2f})")
