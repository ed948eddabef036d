"""Throughput of the CLI path (the training LOOP the shims run: per-step log line with the loss fetched behind an
event, W:951) beside the rate bench.py reports for bare steps: VERDICT r1 item 9.
  python tools/cli_path_rate.py [--steps 200] [--depth 2]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--depth", type=int, default=2)
a = ap.parse_args()
import torch  # noqa: E402
import tethys_speech_amd  # noqa: E402,F401
from tethys_speech_amd import dist as D, train  # noqa: E402
lines = []
t0 = time.time()
model = train.train_whisper(D.DataParallelStrategy(0, 1), model_type="small", batch_size=8, num_batches=a.steps, precision="bf16",
                            device="cuda:0", log=lines.append, loss_fetch_depth=a.depth)
torch.cuda.synchronize()
steps = [l for l in lines if l.startswith("Step ")]
# elapsed time printed on the last and on the 10th line (steady state, model construction excluded)
def elapsed(l):
    return float(l.split("경과: ")[1].split("초")[0])
dt = elapsed(steps[-1]) - elapsed(steps[9])
n = len(steps) - 10
print(f"CLI path (train_whisper loop, per-step log line, loss fetch depth {a.depth}): {dt / n * 1e3:.2f} ms/step = "
      f"{30.0 * 8 * n / dt:.0f} audio-s/s over {n} steps; last line: {steps[-1]}")
