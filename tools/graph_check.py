import sys, os, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import tethys_speech_amd
from tethys_speech_amd import dist as D, optim, train, whisper
from tethys_speech_amd.data import create_dummy_dataset
dev = "cuda:0"
strategy = D.DataParallelStrategy(0, 1)
def run(graph):
    os.environ["TMI_HIP_GRAPH"] = "1" if graph else "0"
    model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
    opt = optim.Adam(1e-4)
    ds = create_dummy_dataset(8, device=dev, rank=0, world=1, seed=1234, drop_remainder=True)
    it = iter(ds)
    first = next(it)
    step = train.make_train_step(strategy, model, opt, first, warmup=2)
    print("graphed" if isinstance(step, train.GraphedTrainStep) else "eager", flush=True)
    losses = []
    for _ in range(5):
        losses.append(float(step(next(it)).item()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        loss = step(next(it))
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"graph={graph}: {dt/40*1e3:.2f} ms/step (host {th/40*1e3:.2f}), losses {['%.4f' % x for x in losses]}, iterations {opt.iterations}", flush=True)
    return losses
a = run(False)
b = run(True)
print("max loss diff", max(abs(x - y) for x, y in zip(a, b)))
