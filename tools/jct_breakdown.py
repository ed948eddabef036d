"""Where the JCT of a short job goes (speech_jobs/whisper_dist.py --batch_size 4 --num_batches 30): model build, dataset, first
step (module load), steady steps, checkpoint, weight save."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.time()
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import dist as D, optim, train, whisper
from tethys_speech_amd.data import create_dummy_dataset
t1 = time.time(); print(f"imports {t1 - t0:.2f} s")
torch.cuda.set_device(0); torch.zeros(1, device="cuda:0"); torch.cuda.synchronize()
t2 = time.time(); print(f"device init {t2 - t1:.2f} s")
model = whisper.create_whisper_model("small", device="cuda:0", precision="bf16", seed=1234); model.refresh_shadows(); torch.cuda.synchronize()
t3 = time.time(); print(f"model build {t3 - t2:.2f} s")
it = iter(create_dummy_dataset(4, device="cuda:0", seed=1234)); torch.cuda.synchronize()
t4 = time.time(); print(f"dataset {t4 - t3:.2f} s")
strat = D.DataParallelStrategy(0, 1); opt = optim.Adam(1e-4); model.enable_dropout(0.1, 0.1, seed=1)
l = train.distributed_train_step(strat, model, next(it), opt); l.item()
t5 = time.time(); print(f"first step {t5 - t4:.2f} s")
for _ in range(29):
    l = train.distributed_train_step(strat, model, next(it), opt)
l.item()
t6 = time.time(); print(f"29 steps {t6 - t5:.2f} s")
os.makedirs("/tmp/ck", exist_ok=True)
train.save_checkpoint(model, opt, "/tmp/ck/a.pt", step=30)
t7 = time.time(); print(f"checkpoint {t7 - t6:.2f} s")
train.save_weights(model, "/tmp/ck/w")
t8 = time.time(); print(f"save_weights {t8 - t7:.2f} s")
