"""Generates the golden fixtures under tests/golden/ from the oracle (run in the build
container; the GPU box only reads the JSON).  The reference itself cannot be run (TensorFlow
absent), so these vectors pin the HIP path to the oracle, not to a live TF.

  python tests/golden/make_golden.py [--tiny-steps 10]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def tiny_curve(steps, seed=1234, dtype=torch.float64):
    cfg = O.make_config("tiny")
    params = O.init_params(cfg, seed=seed, dtype=torch.float32)
    params = {k: v.to(dtype) for k, v in params.items()}
    feats, labels = O.create_dummy_pool(seed=seed)
    t0 = time.time()
    losses, _ = O.train_steps(cfg, params, feats, labels, 2, steps, lr=1e-4)
    return {"model": "whisper-tiny (384/6h/1536/4+4, W:859-865)", "batch_size": 2, "steps": steps, "seed": seed,
            "lr": 1e-4, "oracle_dtype": str(dtype), "losses": losses, "oracle_seconds": time.time() - t0}


def small_ref_curve(steps=10, batch=8, seed=1234, dtype=torch.float64):
    """BASELINE.json's headline model at its headline batch: Whisper small-ref (768/12h/3072/4+4, W:13-18), B = 8, the
    bench.py pool (seed 1234), Adam 1e-4 (W:901), dropout 0.  ``dataset.batch(8).repeat()`` without drop_remainder
    (W:812-815): step 6 is the short batch (samples 48, 49).  ~1 min per step in fp64 on 8 cores, ~15 GB."""
    cfg = O.make_config("small")
    params = {k: v.to(dtype) for k, v in O.init_params(cfg, seed=seed, dtype=torch.float32).items()}
    feats, labels = O.create_dummy_pool(seed=seed)
    t0 = time.time()
    losses, _ = O.train_steps(cfg, params, feats, labels, batch, steps, lr=1e-4)
    return {"model": "whisper-small-ref (768/12h/3072/4+4, W:13-18,888)", "batch_size": batch, "steps": steps, "seed": seed,
            "lr": 1e-4, "oracle_dtype": str(dtype), "losses": losses, "oracle_seconds": time.time() - t0}


def w2v_curve(steps, model_size="base", seed=1234, dtype=torch.float64):
    from oracle import wav2vec2_oracle as V
    cfg = V.make_config(model_size)
    params = {k: v.to(dtype) for k, v in V.init_params(cfg, seed=seed, dtype=torch.float32).items()}
    pool = V.create_dummy_pool(seed=seed)
    t0 = time.time()
    trace = []
    losses, _ = V.train_steps(cfg, params, pool, 2, steps, seed=seed + 1, lr=3e-5, code_trace=trace)
    # ``code_indices`` [steps][B][T][G]: the oracle's quantiser choices, the teacher-forcing input of the bf16 test (the
    # hard argmin is discontinuous: a bf16 run on its OWN choices leaves the golden trajectory at the first flipped code)
    return {"model": f"wav2vec2-{model_size} pretraining (V:24-128), 2 s clips", "batch_size": 2, "steps": steps,
            "seed": seed, "neg_seed": seed + 1, "lr": 3e-5, "oracle_dtype": str(dtype), "losses": losses,
            "code_indices": [t.tolist() for t in trace], "oracle_seconds": time.time() - t0}


def single_curve(steps=10, seed=1234, dtype=torch.float64):
    """BASELINE configs[0] read as the file is named (speech_jobs/whisper_single.py = single-device Wav2Vec2-base):
    batch 2, 10 steps, 5 s clips, Adam 3e-5 / eps 1e-7, roll negatives from default_rng(42)."""
    from oracle import wav2vec2_oracle as V
    cfg = V.make_config("base")
    params = {k: v.to(dtype) for k, v in V.init_params(cfg, seed=seed, dtype=torch.float32).items()}
    pool = V.create_dummy_pool(seed=seed, length=80000)
    t0 = time.time()
    losses, _ = V.train_steps_single(cfg, params, pool, 2, steps, seed=42, lr=3e-5)
    return {"model": "wav2vec2-base pretraining as speech_jobs/whisper_single.py runs it (S:), 5 s clips", "batch_size": 2,
            "steps": steps, "seed": seed, "neg_seed": 42, "lr": 3e-5, "oracle_dtype": str(dtype), "losses": losses,
            "oracle_seconds": time.time() - t0}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny-steps", type=int, default=10)
    ap.add_argument("--only", choices=["whisper", "w2v", "small", "single", "all"], default="all")
    a = ap.parse_args()
    torch.set_num_threads(8)
    if a.only in ("whisper", "all"):
        out = tiny_curve(a.tiny_steps)
        json.dump(out, open(os.path.join(HERE, "whisper_tiny_b2_10steps.json"), "w"), indent=1)
        print(out)
    if a.only in ("small", "all"):
        out = small_ref_curve(10)
        json.dump(out, open(os.path.join(HERE, "whisper_small_ref_b8_10steps.json"), "w"), indent=1)
        print(out)
    if a.only in ("single", "all"):
        out = single_curve(10)
        json.dump(out, open(os.path.join(HERE, "whisper_single_w2v_base_b2_10steps.json"), "w"), indent=1)
        print(out)
    if a.only in ("w2v", "all"):
        out = w2v_curve(5)
        json.dump(out, open(os.path.join(HERE, "wav2vec2_base_b2_5steps.json"), "w"), indent=1)
        print(out)
