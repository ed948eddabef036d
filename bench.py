#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec/node of the Whisper small-ref training step.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = forward + backward + gradient all-reduce (SUM, N > 1) + Adam + bf16 shadow
refresh on one synthetic batch already resident in HBM (BASELINE.json configs[1]: Whisper
"small" as the reference defines it — 768/12h/3072, 4+4 layers, W:13-18 — bf16 compute,
per-GPU batch 8, 30 s clips = [8, 80, 3000] features + [8, 100] labels).  Weak scaling: the
per-GPU batch is fixed, value = 30 s * 8 * N * K / max-over-ranks wall time.
The step runs as the reference's does (training=True): its Dropout layers (W:29-30, rates 0.1 / 0.1) are
active, with counter-based masks (--dropout off gives the rates-0 configuration the loss-curve parity tests pin).

Also reported on the same JSON line:
  roofline      the dominant kernel class (the MFMA GEMM, tmi_gemm): algorithmic FLOPs of
                every tmi_gemm launch in a step / their device time measured with HIP events
                recorded on the launch stream during an instrumented pass of the same steps
                (weight-gradient side stream off, so launches do not overlap), against the
                2.5 PFLOP/s dense bf16 MFMA peak.
  roofline_classes  the same measurement for every instrumented kernel class (GEMM and fused attention
                against the MFMA peak by algorithmic flop; Adam, LayerNorm, bias column sums and the
                cross-entropy against the 8 TB/s HBM peak by algorithmic bytes).
  cpu_baseline  the oracle (restated reference CPU path, TensorFlow unavailable) timed on the
                host cores, rank 0, N = 1 only, on a bounded sample (batch 2, 1 warm-up + up to
                6 timed steps, ~14 s, of the same model and clip length; dropout rates 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CLIP_SECONDS = 30.0
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (guides/MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0            # HBM3E peak (same guide; ~6.3 TB/s is what streams reach)
GF_PER_SAMPLE_BY_SIZE = {"tiny": 141.3, "small": 449.1, "large": 8320.6}
GF_PER_SAMPLE = 449.1           # fwd+bwd algorithmic GFLOP per 30 s sample, small-ref


def cpu_baseline(model_type, seq_len, sample_batch=2, max_steps=6, budget_s=14.0):
    import torch
    from oracle import whisper_oracle as O  # checker, timed as the CPU baseline only
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the 1-GPU box's CPU share; more threads than share only thrash
    torch.set_num_threads(cores)
    cfg = O.make_config(model_type, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    params = O.init_params(cfg, seed=1234, dtype=torch.float32)
    feats, labels = O.create_dummy_pool(seed=1234, seq_len=seq_len, num_samples=sample_batch * (max_steps + 1))
    tw = time.time()
    O.train_steps(cfg, params, feats, labels, sample_batch, 1)  # warm-up
    log(f"cpu baseline warm-up step: {time.time() - tw:.1f} s on {cores} threads")
    # bounded sample: as many timed steps as fit ~budget_s of CPU work (the warm-up step sizes them)
    steps = max(1, min(max_steps, int(budget_s / max(time.time() - tw, 1e-3))))
    t0 = time.time()
    O.train_steps(cfg, params, feats[sample_batch:], labels[sample_batch:], sample_batch, steps)
    dt = time.time() - t0
    return {"value": CLIP_SECONDS * sample_batch * steps / dt, "unit": "audio-seconds/sec", "cores": cores,
            "kind": "port",
            "sample": f"restated reference CPU path (TensorFlow unavailable): oracle fp32, whisper-{model_type}-ref, "
                      f"batch {sample_batch}, {steps} timed steps after 1 warm-up, {dt / steps:.2f} s/step"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def bench_wav2vec2(args, strategy, dev, rank, world):
    """Secondary workload (BASELINE configs[3]): Wav2Vec2-base pre-training step, 2 s clips."""
    import numpy as np
    import torch
    from tethys_speech_amd import ops, optim, train, wav2vec2
    from tethys_speech_amd.data import W2VDummyDataset
    size = "base" if args.model_type == "small" else args.model_type
    model = wav2vec2.create_full_model("pretraining", size, device=dev, precision=args.precision, seed=1234)
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    if args.dropout == "reference":
        c = model.config
        model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=1234 * 1000003 + rank, act_p=c.activation_dropout)
    opt = optim.Adam(learning_rate=3e-5, epsilon=1e-8)
    ds = W2VDummyDataset(args.batch_size, device=dev, rank=rank, world=world, seed=1234)
    it = iter(ds)
    rng = np.random.default_rng(1235)
    negs = [torch.from_numpy(wav2vec2.sample_negative_indices(rng, args.batch_size, 100, 100)).to(dev)
            for _ in range(args.steps + args.warmup)]

    def one_step(i):
        return train.wav2vec2_train_step(strategy, model, next(it), negs[i], opt)

    for i in range(args.warmup):
        one_step(i)
    strategy.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
    torch.cuda.synchronize()
    strategy.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    log(f"wav2vec2-{size}: {dt / args.steps * 1e3:.2f} ms/step, loss {float(loss.item()):.4f}")
    if rank == 0:
        gb = args.batch_size * world
        print(json.dumps({
            "metric": "audio-seconds/sec/node (Wav2Vec2-base pretrain step, 2 s clips)", "value": 2.0 * gb * args.steps / dt,
            "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"wav2vec2-{size} pre-training step (V:), per-GPU batch {args.batch_size}, 2 s clips [32000]",
                       "dropout": "reference rates (0.1 / 0.1 / 0.1)" if args.dropout == "reference" else "off",
                       "global_batch": gb, "parallelism": f"dp{world}", "last_loss": float(loss.item())}}))
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--model_type", default="small")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="whisper", choices=["whisper", "wav2vec2"],
                    help="whisper = BASELINE configs[1] (headline); wav2vec2 = configs[3] model (base, 2 s clips)")
    ap.add_argument("--dropout", choices=["off", "reference"], default="reference",
                    help="reference (default): the reference's training-mode Dropout layers (W:29-30, rates 0.1 / 0.1) are "
                         "active, as in its distributed_train_step (training=True), with counter-based masks; "
                         "off: rates 0, the configuration the loss-curve parity tests pin")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import ops, optim, train, whisper
    from tethys_speech_amd.data import create_dummy_dataset

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world, backend="nccl")
    if args.workload == "wav2vec2":
        return bench_wav2vec2(args, strategy, dev, rank, world)

    model = whisper.create_whisper_model(args.model_type, device=dev, precision=args.precision, seed=1234)
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    if args.dropout == "reference":
        model.enable_dropout(model.config.dropout, model.config.attention_dropout, seed=1234 * 1000003 + rank)
    opt = optim.Adam(learning_rate=1e-4)
    ds = create_dummy_dataset(args.batch_size, device=dev, rank=rank, world=world, seed=1234, drop_remainder=True)
    it = iter(ds)

    def one_step():
        return train.distributed_train_step(strategy, model, next(it), opt)

    log(f"model ready ({model.arena.n_params} params), warming up {args.warmup} steps")
    for i in range(args.warmup):
        tw = time.perf_counter()
        one_step()
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms")
    strategy.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    t_host = time.perf_counter() - t0  # host time to enqueue the steps (no sync inside a step)
    torch.cuda.synchronize()
    strategy.barrier()
    dt = time.perf_counter() - t0
    log(f"host enqueue {t_host / args.steps * 1e3:.2f} ms/step of {dt / args.steps * 1e3:.2f} ms/step wall")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    last_loss = float(loss.item())
    log(f"timed {args.steps} steps: {dt / args.steps * 1e3:.2f} ms/step, loss {last_loss:.4f}")

    roof = None
    if not args.no_roofline:
        # instrumented pass: HIP events around every tmi_gemm launch, on the launch stream.  The
        # weight-gradient side stream is switched off for it, so each GEMM runs alone and its
        # duration is its own (in the timed region above weight gradients overlap the dgrad chain).
        overlap = model._side is not None
        model.enable_wgrad_stream(False)
        # One step per probe object, read out before the next: with more than ~500 timing events
        # outstanding the runtime stalls the stream for ~50 ms on a record, which is not GEMM time.
        nprof = min(args.steps, 3)
        ops.PROFILE = ops.OpProfile()
        for _ in range(nprof):
            one_step()
            ops.PROFILE.flush()
        prof, ops.PROFILE = ops.PROFILE, None
        ms, flops, launches = prof.totals("gemm")
        model.enable_wgrad_stream(overlap)
        # the other kernel classes, each against the roof that bounds it ("report per kernel class")
        classes = []
        for cls, bound, peak, unit, scale in (("gemm", "mfma", MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12),
                                              ("attention", "mfma", MFMA_BF16_PEAK_TFLOPS, "TFLOP/s", 1e12),
                                              ("adam", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                              ("layernorm", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                              ("colsum", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                              ("xent", "hbm", HBM_PEAK_GBS, "GB/s", 1e9)):
            cm, cw, cn = prof.totals(cls)
            if cn == 0 or cm <= 0:
                continue
            ach = cw / (cm * 1e-3) / scale
            classes.append({"class": cls, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                            "ms_per_step": cm / nprof, "launches_per_step": cn / nprof})
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM traffic cannot be counted from inside the process: it comes from the last committed PMC
        # passes over this same command (tools/pmc_traffic.py), bytes per tmi_gemm launch
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_gemm_pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic, traffic_src = pmc["hbm_bytes_per_launch"], "profiles/r01_gemm_pmc_traffic.json: " + pmc["source"]
        except Exception:
            pass
        roof = {"bound": "mfma", "kernel": "tmi_gemm kernels (gemm_fast_kernel / gemm_p8_kernel / gemm_kernel: all Dense/Conv1D fwd, dgrad, wgrad GEMMs)",
                "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src,
                "launches_per_step": launches / nprof, "gemm_ms_per_step": ms / nprof,
                "gemm_gflop_per_step": flops / nprof / 1e9,
                "avg_launch_us": ms * 1e3 / max(1, launches)}

    if rank == 0:
        gb = args.batch_size * world
        value = CLIP_SECONDS * gb * args.steps / dt
        c = model.config
        shape_note = f"{c.d_model}/{c.encoder_attention_heads}h/{c.d_ff}, {c.encoder_layers}+{c.decoder_layers} layers"
        gf_sample = GF_PER_SAMPLE_BY_SIZE.get(args.model_type)
        out = {
            "metric": f"audio-seconds/sec/node (Whisper-{args.model_type}, 30 s clips)",
            "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"whisper-{args.model_type}-ref (reference '{args.model_type}': {shape_note}) "
                                   f"train step, per-GPU batch {args.batch_size}, 30 s clips [80x3000], S=100",
                       "global_batch": gb, "parallelism": f"dp{world}", "last_loss": last_loss,
                       "dropout": ("reference rates (0.1 / 0.1), counter-based masks" if args.dropout == "reference"
                                   else "off (rates 0: the configuration the parity tests pin)"),
                       "step_tflops": gf_sample * gb * args.steps / dt / 1e3 if gf_sample else None},
        }
        if roof is not None:
            out["roofline"] = roof
            out["roofline_classes"] = classes
        if world == 1 and not args.no_cpu_baseline:
            log("timing the restated reference CPU path (oracle) on the host cores")
            out["cpu_baseline"] = cpu_baseline(args.model_type, 3000)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
