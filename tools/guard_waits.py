"""Which buffers make the main stream wait for the weight-gradient stream in one Whisper small-ref step: every
KernelBlocks._guard_write / _wait_events call that actually enqueues a stream wait, with the workspace names of the
tensors involved (host-side bookkeeping only; pair with the gaps of the overlapped rocprofv3 timeline)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import blocks, dist as D, optim, train, whisper
from tethys_speech_amd.data import create_dummy_dataset

dev = "cuda:0"
strategy = D.DataParallelStrategy(0, 1)
model = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=1234)
model.enable_dropout(0.1, 0.1, seed=1)
opt = optim.Adam(1e-4)
it = iter(create_dummy_dataset(8, device=dev, rank=0, world=1, seed=1234, drop_remainder=True))
train.USE_PLAN = False
for _ in range(3):
    train.distributed_train_step(strategy, model, next(it), opt, pipelined=True)
torch.cuda.synchronize()
names = {}
for k, t in model.ws.items():
    if torch.is_tensor(t):
        names.setdefault(t.data_ptr(), []).append(k)
log = []
orig_guard, orig_wait = blocks.KernelBlocks._guard_write, blocks.KernelBlocks._wait_events


def guard(self, *tensors):
    if self._side is not None and self._side_reads:
        for t in tensors:
            if t.data_ptr() in self._side_reads:
                import traceback
                fr = traceback.extract_stack(limit=4)[0:3]
                log.append(("guard_write", names.get(t.data_ptr(), ["?"])[:2], [f"{f.name}:{f.lineno}" for f in fr]))
    return orig_guard(self, *tensors)


def wait(self, events):
    events = list(events)
    if events:
        import traceback
        fr = traceback.extract_stack(limit=4)[0:3]
        log.append(("wait_events", len(events), [f"{f.name}:{f.lineno}" for f in fr]))
    return orig_wait(self, events)


blocks.KernelBlocks._guard_write, blocks.KernelBlocks._wait_events = guard, wait
train.distributed_train_step(strategy, model, next(it), opt, pipelined=True)
torch.cuda.synchronize()
for e in log:
    print(e)
