"""Host time to ENQUEUE one step (device idle at the start of each: the launch queue never fills, so this is pure host
work: Python + ctypes + HIP runtime), beside the device time of the same step - issued from Python launch by launch
(eager) and replayed from a launch plan (tethys_speech_amd/plan.py).  usage: host_step_time.py [whisper|wav2vec2] [replicas]
``replicas``: the job's N > 1 step on this one GPU - a one-rank RCCL group with the strategy's world-1 short-circuits off
(``force_collectives``): bucketed all-reduces issued from inside backward, early Adam slices per bucket; in the replayed
step the collectives and their waits are callback nodes of the plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import dist as D, optim, train, wav2vec2, whisper
from tethys_speech_amd.data import W2VDummyDataset, create_dummy_dataset

which = sys.argv[1] if len(sys.argv) > 1 else "whisper"
dev = "cuda:0"
replicas = "replicas" in sys.argv[2:]
if replicas:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    torch.cuda.set_device(0)
    strategy = D.DataParallelStrategy(0, 1, backend="nccl", force_collectives=True)
else:
    strategy = D.DataParallelStrategy(0, 1)
if which == "whisper":
    model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
    model.enable_dropout(0.1, 0.1, seed=1)
    opt = optim.Adam(1e-4)
    it = iter(create_dummy_dataset(8, device=dev, rank=0, world=1, seed=1234, drop_remainder=True))
    batch = lambda i: next(it)
    kind, label = "whisper", "whisper small-ref"
else:
    model = wav2vec2.create_full_model("pretraining", "base", device=dev, precision="bf16", seed=1234)
    c = model.config
    model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=1, act_p=c.activation_dropout)
    opt = optim.Adam(3e-5, epsilon=1e-8)
    it = iter(W2VDummyDataset(8, device=dev, seed=1234))
    rng = np.random.default_rng(1)
    negs = [torch.from_numpy(wav2vec2.sample_negative_indices(rng, 8, 100, 100)).to(dev) for _ in range(8)]
    batch = lambda i: (next(it), negs[i % 8])
    kind, label = "wav2vec2", "wav2vec2-base"
for planned in (False, True):
    train.USE_PLAN = planned
    step = train.planned_step(strategy, model, opt, kind, pipelined=True)
    for i in range(6):
        step(*batch(i))
    torch.cuda.synchronize()
    host, total = [], []
    for i in range(20):
        b = batch(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(*b)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    host.sort(); total.sort()
    how = "launch plan replay" if planned else "eager (one ctypes call per launch)"
    extra = ""
    if planned and step.planned is not None:
        pl = [v["plan"] for v in step.planned._by_sig.values() if v.get("plan") is not None]
        extra = f"; plan: {pl[0].launches} launches, {pl[0].nodes} nodes ({pl[0].callbacks} host callbacks), {step.planned.replays} replays"
    if replicas:
        label = label.split(" [")[0] + " [one-rank RCCL group, collectives forced]"
    print(f"{label} step from an idle device, {how}: host enqueue median {host[10]:.2f} ms (min {host[0]:.2f}), "
          f"enqueue + drain median {total[10]:.2f} ms (min {total[0]:.2f}){extra}")
