"""bench.py end to end on the GPU (a short run): ONE JSON line on stdout with the keys the driver's contract names, the
roofline object measured live (HIP events on the launch stream) and the bounded CPU baseline with its loss cross-check."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config"}


def _run(args, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout: the JSON (progress goes to stderr)
    return json.loads(lines[0])


def test_bench_line_whisper_contract(dev):
    d = _run(["--steps", "6", "--warmup", "2", "--cpu-budget", "5"])
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert d["unit"] == "audio-seconds/sec" and "Whisper-small" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 30.0 * 8 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert c["loss_check"]["agree"] is True


def test_bench_line_wav2vec2_contract(dev):
    d = _run(["--workload", "wav2vec2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"])
    assert KEYS <= set(d) and "roofline" in d and "roofline_classes" in d
    assert "Wav2Vec2-base" in d["metric"] and d["roofline"]["bound"] == "mfma"
    assert abs(d["value"] - 2.0 * 8 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-6 * d["value"]


def test_bench_plain_command_self_launches_two_ranks(dev):
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own ranks (child
    process, torch.distributed.run) instead of exiting.  Rehearsed on the one card of this box: gloo as the exchange
    backend, both ranks on device 0.  The line must be a valid N = 2 line with the exchange diagnostics."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TETHYS_DIST_BACKEND="gloo", TETHYS_ONE_DEVICE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--no-roofline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert KEYS <= set(d) and d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 30.0 * 16 * 4 / (d["ms_per_step"] * 4e-3)) <= 1e-6 * d["value"]
    c = d["config"]
    assert c["rccl_ranks"] == 2 and c["exposed_exchange_ms"] is not None and len(c["host_enqueue_ms"]) == 2
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
