#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec/node of the Whisper small-ref training step.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...; the plain command
   with N > 1 and no WORLD_SIZE starts exactly that as a child process and relays rank 0's line: self_launch)

A step = forward + backward + gradient all-reduce (SUM, N > 1) + Adam + bf16 shadow
refresh on one synthetic batch already resident in HBM (BASELINE.json configs[1]: Whisper
"small" as the reference defines it — 768/12h/3072, 4+4 layers, W:13-18 — bf16 compute,
per-GPU batch 8, 30 s clips = [8, 80, 3000] features + [8, 100] labels).  Weak scaling: the
per-GPU batch is fixed, value = 30 s * 8 * N * K / max-over-ranks wall time.
The step runs as the reference's does (training=True): its Dropout layers (W:29-30, rates 0.1 / 0.1) are
active, with counter-based masks (--dropout off gives the rates-0 configuration the loss-curve parity tests pin).
``--workload wav2vec2`` times BASELINE configs[3]'s model (Wav2Vec2-base pre-training step, 2 s clips) the same way.

Every line carries, besides the contract's fields:
  roofline      the dominant kernel class (the MFMA GEMM, tmi_gemm): algorithmic FLOPs of
                every tmi_gemm launch in a step / their device time measured with HIP events
                recorded on the launch stream during an instrumented pass of the same steps
                (weight-gradient side stream off, so launches do not overlap), against the
                2.5 PFLOP/s dense bf16 MFMA peak (157.3 TFLOP/s fp32 MFMA with --precision fp32).
  roofline_classes  the same measurement for every instrumented kernel class (GEMM and fused attention
                against the MFMA peak by algorithmic flop — attention: forward 4·B·H·Tq·Tk·64, backward twice that,
                SURVEY 8d; Adam, LayerNorm, GroupNorm, bias column sums and the cross-entropy against the 8 TB/s
                HBM peak by algorithmic bytes).
  cpu_baseline  the oracle (restated reference CPU path, TensorFlow unavailable) timed on the host cores, rank 0,
                N = 1 only, on a bounded sample of the SAME workload (same model, per-GPU batch, clip length, seed;
                dropout rates 0): thread setting 1 = the physical cores of one socket this process may use,
                setting 2 = 5 threads (the reference pod's CPU limit, sample_tfjobs/whisper-dist.yaml:36); the
                CPU's per-step losses are compared with the GPU fp32 path on the same steps.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3    # exact-fp32 MFMA = the fp32 vector rate (same guide)
HBM_PEAK_GBS = 8000.0            # HBM3E peak (same guide; ~6.3 TB/s is what streams reach)
GF_PER_SAMPLE_BY_SIZE = {"tiny": 141.3, "small": 449.1, "large": 8320.6}  # SURVEY.md 8(d), 30 s clips
W2V_GF_PER_SAMPLE = {"base": 83.43}                                        # SURVEY.md 8(d), 2 s clips
POD_CPU_LIMIT = 5                # sample_tfjobs/whisper-dist.yaml:36
CPU_TIMED_MIN = 3                # 1 warm-up + >= 3 timed steps per thread setting


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------ CPU baseline
def host_cpu():
    """(model string, threads for 'all physical cores of one socket this process may use', nproc)."""
    model, per_socket = "unknown", None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            if line.startswith("cpu cores") and per_socket is None:
                per_socket = int(line.split(":", 1)[1])
    except Exception:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    # "all physical cores of one socket" (and 5, the reference pod's limit) - of the cores this
    # process may actually USE: a container's CPU quota (cgroup cpu.max / cfs_quota) is honoured, because threads beyond it
    # only time-slice (measured on a 16-core share of a 64-core EPYC 9575F: 9-11 s/step on 16 threads, 15.8 s/step on 64).
    # TMI_BENCH_CPU_SHARE overrides the quota (0 / unset: read it from the cgroup).
    share = int(os.environ.get("TMI_BENCH_CPU_SHARE", "0")) or cgroup_cpu_quota() or (per_socket or avail)
    threads = max(1, min(avail, per_socket or avail, share))
    return model, threads, avail


def cgroup_cpu_quota():
    """CPUs this process's cgroup may use (ceil of quota / period), or 0 when there is no quota."""
    import math
    try:  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, math.ceil(int(q) / int(p)))
    except Exception:
        pass
    try:  # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, math.ceil(q / p))
    except Exception:
        pass
    return 0


def _time_oracle(run_steps, threads, warm, timed):
    """run_steps(n) advances the oracle n steps and returns their losses."""
    import torch
    torch.set_num_threads(threads)
    losses = []
    if warm:
        t0 = time.time()
        losses += run_steps(warm)
        log(f"cpu baseline ({threads} threads) warm-up: {(time.time() - t0) / warm:.1f} s/step")
    t0 = time.time()
    losses += run_steps(timed)
    return (time.time() - t0) / timed, losses


def cpu_baseline_whisper(model_type, batch, dev, budget_s):
    """Oracle fp32 on the bench batch (pool seed 1234, batch ``batch``, 30 s clips).  Setting 1 (all physical cores of one
    socket): 1 warm-up + at least 3 timed steps (more if they fit ~budget_s, at most 6); setting 2 (5 threads): continues
    from there for 3 timed steps (no new warm-up: the CPU path compiles nothing).  BASELINE.md section 3.4: 1 warm-up +
    >= 3 timed steps.  The GPU fp32 path then runs the same steps from the same initial weights."""
    import numpy as np
    import torch
    from oracle import whisper_oracle as O  # checker, timed as the CPU baseline only
    cpu_model, threads, nproc = host_cpu()
    cfg = O.make_config(model_type, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    params = O.init_params(cfg, seed=1234, dtype=torch.float32)
    params0 = {k: v.clone() for k, v in params.items()}
    feats, labels = O.create_dummy_pool(seed=1234)
    it = O.batches(feats, labels, batch)
    state = O.AdamState()
    used = []

    def run_steps(n):
        out = []
        for _ in range(n):
            f, l = next(it)
            used.append((f, l))
            loss, g = O.loss_and_grads(params, torch.from_numpy(np.ascontiguousarray(f)), torch.from_numpy(np.ascontiguousarray(l)), cfg)
            O.adam_step(params, g, state, lr=1e-4)
            out.append(float(loss))
        return out

    t0 = time.time()
    first = run_steps(1)
    warm_s = time.time() - t0
    log(f"cpu baseline warm-up step: {warm_s:.1f} s on {threads} threads ({cpu_model})")
    torch.set_num_threads(threads)
    n1 = max(CPU_TIMED_MIN, min(6, int(budget_s / max(warm_s, 1e-3))))
    s1, l1 = _time_oracle(run_steps, threads, 0, n1)
    log(f"cpu baseline: {s1:.2f} s/step over {n1} timed steps on {threads} threads")
    s5, l5 = _time_oracle(run_steps, min(POD_CPU_LIMIT, nproc), 0, CPU_TIMED_MIN)
    log(f"cpu baseline: {s5:.2f} s/step over {CPU_TIMED_MIN} timed steps on {min(POD_CPU_LIMIT, nproc)} threads")
    cpu_losses = first + l1 + l5
    # the GPU fp32 (parity) path on the same batches from the same initial weights
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, whisper
    check = {"cpu_losses": cpu_losses, "tolerance": 1e-3}
    try:
        m = whisper.create_whisper_model(model_type, device=dev, precision="fp32", seed=1234)
        m.arena.load_ref(params0)
        opt = optim.Adam(learning_rate=1e-4)
        strat = D.DataParallelStrategy(0, 1)
        dev_batches = [(torch.from_numpy(np.ascontiguousarray(f)).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev))
                       for f, l in used]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl_t = [train.distributed_train_step(strat, m, b_, opt).clone() for b_ in dev_batches]
        torch.cuda.synchronize()
        fp32_ms = (time.perf_counter() - t0) / len(dev_batches) * 1e3  # (the first step includes the workspace set-up)
        gl = [float(t.item()) for t in gl_t]
        diff = max(abs(a - b) for a, b in zip(gl, cpu_losses))
        check.update({"gpu_fp32_losses": gl, "max_abs_diff": diff, "agree": diff <= 1e-3,
                      "gpu_fp32_ms_per_step": fp32_ms,
                      "gpu_fp32_note": "the parity mode (exact-fp32 MFMA, scores materialised as the reference does), timed over these steps"})
        log(f"loss cross-check CPU oracle fp32 vs GPU fp32 over {len(gl)} steps: max |d| = {diff:.2e}")
        del m
        torch.cuda.empty_cache()
    except Exception as e:  # the cross-check must not lose the bench line
        check.update({"error": f"{type(e).__name__}: {e}"})
    clip = 30.0
    return {"value": clip * batch / s1, "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
            "sample": f"restated reference CPU path (TensorFlow unavailable): oracle fp32, whisper-{model_type}-ref, batch {batch}, "
                      f"30 s clips, pool seed 1234, 1 warm-up + {n1} timed steps, {s1:.2f} s/step on {threads} threads "
                      f"(the physical cores of one socket this process may use: cgroup quota {cgroup_cpu_quota() or 'none'}; "
                      f"{CPU_TIMED_MIN} more timed steps on {min(POD_CPU_LIMIT, nproc)} threads: {s5:.2f} s/step)",
            "cpu_model": cpu_model, "nproc": nproc, "cgroup_cpu_quota": cgroup_cpu_quota(),
            "settings": [{"threads": threads, "s_per_step": s1, "value": clip * batch / s1, "timed_steps": n1,
                          "note": "all physical cores of one socket this process may use (BASELINE.md 3.3; cgroup quota honoured)"},
                         {"threads": min(POD_CPU_LIMIT, nproc), "s_per_step": s5, "value": clip * batch / s5, "timed_steps": CPU_TIMED_MIN,
                          "note": "the reference pod's CPU limit (sample_tfjobs/whisper-dist.yaml:36)"}],
            "loss_check": check}


def cpu_baseline_w2v(size, batch, dev, budget_s, single=False):
    """``single``: speech_jobs/whisper_single.py's step (S:) - 5 s clips, roll-based negatives, nothing clipped, Adam eps 1e-7."""
    import numpy as np
    import torch
    from oracle import wav2vec2_oracle as V
    cpu_model, threads, nproc = host_cpu()
    cfg = V.make_config(size)
    params = V.init_params(cfg, seed=1234, dtype=torch.float32)
    params0 = {k: v.clone() for k, v in params.items()}
    pool = V.create_dummy_pool(seed=1234, length=80000) if single else V.create_dummy_pool(seed=1234)
    T = V.feature_lengths(cfg, pool.shape[1])[-1]
    rng = np.random.default_rng(1235)
    it = V.batches(pool, batch)
    state = V.AdamState()
    used = []

    def run_steps(n):
        out = []
        for _ in range(n):
            a = next(it)
            if single:
                neg = V.sample_negative_indices_roll(rng, T, cfg.num_negatives)
                used.append((a, neg))
                loss, g, _ = V.loss_and_grads_single(params, torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(neg), cfg)
                V.adam_step(params, g, state, lr=3e-5, eps=1e-7)
            else:
                neg = V.sample_negative_indices(rng, batch, T, cfg.num_negatives)
                used.append((a, neg))
                loss, g, _ = V.loss_and_grads(params, torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(neg), cfg, 1)
                g, _ = V.clip_by_global_norm(g, 1.0)
                g = V.clip_by_norm_each(g, 1.0)
                V.adam_step(params, g, state, lr=3e-5)
            out.append(float(loss))
        return out

    torch.set_num_threads(threads)
    t0 = time.time()
    first = run_steps(1)
    warm_s = time.time() - t0
    n1 = max(CPU_TIMED_MIN, min(6, int(budget_s / max(warm_s, 1e-3))))
    s1, l1 = _time_oracle(run_steps, threads, 0, n1)
    s5, l5 = _time_oracle(run_steps, min(POD_CPU_LIMIT, nproc), 0, CPU_TIMED_MIN)
    cpu_losses = first + l1 + l5
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, optim, train, wav2vec2
    check = {"cpu_losses": cpu_losses, "tolerance": "2e-3 relative (the loss is O(400): unnormalised logits / 0.1)"}
    try:
        m = wav2vec2.create_full_model("pretraining", size, device=dev, precision="fp32", seed=1234)
        m.arena.load_ref(params0)
        m.refresh_shadows()
        opt = optim.Adam(learning_rate=3e-5) if single else optim.Adam(learning_rate=3e-5, epsilon=1e-8)
        strat = D.DataParallelStrategy(0, 1)
        if single:
            gl = [float(train.single_train_step(m, torch.from_numpy(np.ascontiguousarray(a)).to(dev),
                                                torch.from_numpy(neg).to(dev), opt).item()) for a, neg in used]
        else:
            gl = [float(train.wav2vec2_train_step(strat, m, torch.from_numpy(np.ascontiguousarray(a)).to(dev),
                                                  torch.from_numpy(neg).to(dev), opt).item()) for a, neg in used]
        rel = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(gl, cpu_losses))
        check.update({"gpu_fp32_losses": gl, "max_rel_diff": rel, "agree": rel <= 2e-3})
        log(f"loss cross-check CPU oracle fp32 vs GPU fp32 over {len(gl)} steps: max rel = {rel:.2e}")
        del m
        torch.cuda.empty_cache()
    except Exception as e:
        check.update({"error": f"{type(e).__name__}: {e}"})
    clip = 5.0 if single else 2.0
    return {"value": clip * batch / s1, "unit": "audio-seconds/sec", "cores": threads, "kind": "port",
            "sample": f"restated reference CPU path (TensorFlow unavailable): oracle fp32, wav2vec2-{size} pre-training step "
                      f"({'S: whisper_single.py' if single else 'V:'}), "
                      f"batch {batch}, {clip:g} s clips, pool seed 1234, 1 warm-up + {n1} timed steps, {s1:.2f} s/step on {threads} threads "
                      f"(the physical cores of one socket this process may use: cgroup quota {cgroup_cpu_quota() or 'none'}; "
                      f"{CPU_TIMED_MIN} more timed steps on {min(POD_CPU_LIMIT, nproc)} threads: {s5:.2f} s/step)",
            "cpu_model": cpu_model, "nproc": nproc, "cgroup_cpu_quota": cgroup_cpu_quota(),
            "settings": [{"threads": threads, "s_per_step": s1, "value": clip * batch / s1, "timed_steps": n1,
                          "note": "all physical cores of one socket this process may use (BASELINE.md 3.3; cgroup quota honoured)"},
                         {"threads": min(POD_CPU_LIMIT, nproc), "s_per_step": s5, "value": clip * batch / s5, "timed_steps": CPU_TIMED_MIN,
                          "note": "the reference pod's CPU limit (sample_tfjobs/wav2vec2-dist.yaml)"}],
            "loss_check": check}


# ------------------------------------------------------------------------------------ roofline probe
def measure_roofline(model, one_step, nprof, precision, tag):
    """Instrumented pass: HIP events around every wrapped launch, on the launch stream.  The weight-gradient side
    stream is switched off for it, so each kernel runs alone and its duration is its own (in the timed region
    weight gradients overlap the dgrad chain)."""
    from tethys_speech_amd import ops, train
    overlap = model._side is not None
    model.enable_wgrad_stream(False)
    adam_overlap, train.ADAM_UNDER_BACKWARD = train.ADAM_UNDER_BACKWARD, False  # the optimizer as one launch, alone
    ops.PROFILE = ops.OpProfile()
    for _ in range(nprof):
        one_step()
        ops.PROFILE.flush()  # one step per read-out: many outstanding timing events stall the stream
    prof, ops.PROFILE = ops.PROFILE, None
    model.enable_wgrad_stream(overlap)
    train.ADAM_UNDER_BACKWARD = adam_overlap
    mfma_peak = MFMA_BF16_PEAK_TFLOPS if precision == "bf16" else MFMA_F32_PEAK_TFLOPS
    classes = []
    for cls, bound, peak, unit, scale in (("gemm", "mfma", mfma_peak, "TFLOP/s", 1e12),
                                          ("attention", "mfma", mfma_peak, "TFLOP/s", 1e12),
                                          ("adam", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                          ("layernorm", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                          ("groupnorm", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                          ("colsum", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                          ("dropout", "hbm", HBM_PEAK_GBS, "GB/s", 1e9),
                                          ("xent", "hbm", HBM_PEAK_GBS, "GB/s", 1e9)):
        cm, cw, cn = prof.totals(cls)
        if cn == 0 or cm <= 0:
            continue
        ach = cw / (cm * 1e-3) / scale
        row = {"class": cls, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
               "ms_per_step": cm / nprof, "launches_per_step": cn / nprof}
        if cls == "adam":
            # "achieved" = bytes the dense launches really move / their time.  The row-sparse embedding-table launch skips idle
            # rows, so its ALGORITHMIC bytes (28 B/param) are not bytes moved: they only enter the second figure.
            rm, rw, rn = prof.totals("adam_rows")
            row["note"] = "achieved = bytes moved by the dense launches / their time; algorithmic_* adds the row-sparse table launch at 28 B/param"
            row["algorithmic_gb_s"] = (cw + rw) / ((cm + rm) * 1e-3) / scale
            row["algorithmic_frac"] = row["algorithmic_gb_s"] / peak
            row["ms_per_step"] = (cm + rm) / nprof
            row["launches_per_step"] = (cn + rn) / nprof
        classes.append(row)
    ms, flops, launches = prof.totals("gemm")
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    # Memory traffic cannot be counted from inside the process: it comes from the last committed PMC passes over this
    # same command (tools/pmc_traffic.py), bytes per tmi_gemm launch.  FETCH_SIZE / WRITE_SIZE are the L2's fabric-side
    # counters (Infinity-Cache hits included, guide section HBM), FETCH_SIZE doubled per the gfx950 note.
    traffic, traffic_src = None, None
    for name in (f"r05_{tag}_gemm_pmc_traffic.json", f"r04_{tag}_gemm_pmc_traffic.json", f"r03_{tag}_gemm_pmc_traffic.json", f"r02_{tag}_gemm_pmc_traffic.json",
                 f"r01_{tag}_gemm_pmc_traffic.json" if tag != "whisper" else "r01_gemm_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pmc = json.load(f)
            traffic = pmc.get("fabric_bytes_per_launch", pmc.get("hbm_bytes_per_launch"))
            traffic_src = f"profiles/{name}: " + pmc["source"]
            break
        except Exception:
            continue
    roof = {"bound": "mfma", "kernel": "tmi_gemm kernels (every Dense / Conv1D forward, dgrad and wgrad GEMM of the step)",
            "achieved": achieved, "peak": mfma_peak, "unit": "TFLOP/s", "frac": achieved / mfma_peak,
            "traffic": traffic,
            "traffic_unit": "L2 fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are counted, so an upper bound on HBM bytes)",
            "traffic_source": traffic_src, "launches_per_step": launches / nprof, "gemm_ms_per_step": ms / nprof,
            "gemm_gflop_per_step": flops / nprof / 1e9, "avg_launch_us": ms * 1e3 / max(1, launches)}
    return roof, classes


def timed_region(one_step, args, strategy, dev, world):
    """The contract's timing: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier + device
    synchronize on both sides; returns the MAX over ranks of the wall time (every rank gets the same number)."""
    import torch
    on_gpu = str(dev).startswith("cuda")
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    for i in range(args.warmup):
        tw = time.perf_counter()
        one_step()
        sync()
        if i < 3:
            log(f"warm-up step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms")
    strategy.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    t_host = time.perf_counter() - t0  # host time to enqueue the steps (no sync inside a step)
    sync()
    strategy.barrier()
    dt = time.perf_counter() - t0
    log(f"host enqueue {t_host / args.steps * 1e3:.2f} ms/step of {dt / args.steps * 1e3:.2f} ms/step wall")
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    return float(tmax.item()), float(loss.item()), t_host / args.steps * 1e3


def exchange_diagnostics(without_exchange, args, strategy, dev, world, ms_with, host_ms):
    """What makes an N > 1 line diagnosable (every rank calls this; collectives inside):
      rccl_ranks            ranks that really took part: an all-reduce of ones over the job's backend
      exposed_exchange_ms   ms/step with the gradient exchange - ms/step of the SAME ranks with it left out
                            (``strategy.exchange_off``; MAX over ranks, same barrier / synchronize bracket): what the
                            exchange costs beyond the backward it hides under
      host_enqueue_ms       per rank: host time to enqueue one step (launch-bound ranks show up here)"""
    import torch
    import torch.distributed as td
    ones = torch.ones(1, dtype=torch.float32, device=dev)
    td.all_reduce(ones)
    k = max(5, min(args.steps, 30))
    sub = argparse.Namespace(steps=k, warmup=3)   # (a fresh launch plan: two eager steps + the recorded one)
    strategy.exchange_off = True
    try:
        dt_off, _, _ = without_exchange(lambda step: timed_region(step, sub, strategy, dev, world))
    finally:
        strategy.exchange_off = False
    hosts = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    td.all_gather(hosts, torch.tensor([host_ms], dtype=torch.float64, device=dev))
    ms_off = dt_off / k * 1e3
    return {"rccl_ranks": int(round(float(ones.item()))), "backend": td.get_backend(),
            "ms_per_step_without_exchange": ms_off, "exposed_exchange_ms": ms_with - ms_off,
            "host_enqueue_ms": [float(h.item()) for h in hosts]}


def pool_size(global_batch):
    """Clips in the synthetic pool: the reference's 50 (W:792, V:1130) unless the global batch needs more."""
    return max(50, 2 * global_batch)


def throughput(clip_seconds, per_gpu_batch, world, steps, dt):
    """Whole-job audio-seconds/sec: weak scaling, the per-GPU batch is fixed as N grows."""
    return clip_seconds * per_gpu_batch * world * steps / dt


def self_launch(gpus, script=None, argv=None):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks ourselves.

    This process has made no GPU call (torch is not even imported yet), so it may start children; it never exec()s.  The
    ranks are `python -m torch.distributed.run --nproc-per-node N bench.py <same argv>` in a CHILD process: their stderr
    is inherited (progress lines stream through), their stdout is relayed line by line and rank 0's JSON line is printed
    again as this process's LAST stdout line; the return code is the child's."""
    import socket
    import subprocess
    with socket.socket() as s:  # a free rendezvous port on the loopback (the container hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)]
    cmd += sys.argv[1:] if argv is None else list(argv)
    log(f"--gpus {gpus} without WORLD_SIZE: starting the ranks as a child process: {' '.join(cmd)}")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    line = None
    for out in child.stdout:
        out = out.rstrip("\n")
        # rank 0's JSON object, wherever it sits in the line: the ranks share one pipe, and another rank's (or a library's)
        # output can land in the middle of a line of rank 0's
        at = out.find('{"metric"')
        obj = None
        if at >= 0:
            try:
                obj, end = json.JSONDecoder().raw_decode(out[at:])
            except ValueError:
                obj = None
        if isinstance(obj, dict):
            line = json.dumps(obj)  # held back: it must be the last line of OUR stdout
            rest = (out[:at] + out[at + end:]).strip()
            if rest:
                print(rest, flush=True)
        else:
            print(out, flush=True)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        log("the ranks exited 0 without a JSON line")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default: > 2 s of timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=8, help="per-GPU batch")
    ap.add_argument("--model_type", default="small")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--workload", default="whisper", choices=["whisper", "wav2vec2", "whisper_single"],
                    help="whisper = BASELINE configs[1] (headline); wav2vec2 = configs[3] model (base, 2 s clips); whisper_single = "
                         "configs[0] as the file is named: speech_jobs/whisper_single.py's single-device Wav2Vec2-base step, 5 s clips")
    ap.add_argument("--dropout", choices=["off", "reference"], default=None,
                    help="reference (default on bf16): the reference's training-mode Dropout layers (W:29-30, rates 0.1 / 0.1) are "
                         "active, as in its distributed_train_step (training=True), with counter-based masks; "
                         "off: rates 0, the configuration the loss-curve parity tests pin (and the fp32 path's only mode)")
    ap.add_argument("--grad_dtype", choices=["fp32", "bf16"], default=os.environ.get("TMI_GRAD_DTYPE", "fp32"),
                    help="wire dtype of the gradient exchange (N > 1)")
    ap.add_argument("--exchange", choices=["allreduce", "rs_ag", "mesh"], default=os.environ.get("TMI_EXCHANGE", "allreduce"),
                    help="form of the gradient exchange (N > 1): dist.DataParallelStrategy")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-plan", action="store_true", help="issue every step from Python instead of replaying a launch plan (tethys_speech_amd/plan.py)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of timed CPU work for the first thread setting")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.dropout is None:
        args.dropout = "reference" if args.precision == "bf16" else "off"

    import numpy as np
    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from tethys_speech_amd import optim, train

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("TETHYS_ONE_DEVICE"):  # rehearsal of the N > 1 path on a one-GPU box (with TETHYS_DIST_BACKEND=gloo)
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    strategy = D.DataParallelStrategy(rank, world, backend=os.environ.get("TETHYS_DIST_BACKEND", "nccl"),
                                      grad_dtype=args.grad_dtype, exchange=args.exchange)
    drop_note = ("reference rates, counter-based masks" if args.dropout == "reference"
                 else "off (rates 0: the configuration the parity tests pin)")
    dtype_name = "bf16" if args.precision == "bf16" else "f32"

    if args.workload == "whisper":
        from tethys_speech_amd import whisper
        from tethys_speech_amd.data import create_dummy_dataset
        clip = 30.0
        model = whisper.create_whisper_model(args.model_type, device=dev, precision=args.precision, seed=1234)
        strategy.broadcast_parameters(model.arena.p)
        model.refresh_shadows()
        if args.dropout == "reference":
            model.enable_dropout(model.config.dropout, model.config.attention_dropout, seed=1234 * 1000003 + rank)
        opt = optim.Adam(learning_rate=1e-4)
        # the reference's pool has 50 clips; weak scaling needs every rank to draw its full batch every step, so the pool grows
        # with the global batch beyond two of them (N = 4: 64 clips, N = 8: 128); N = 1, 2 keep the 50
        it = iter(create_dummy_dataset(args.batch_size, device=dev, rank=rank, world=world, seed=1234, drop_remainder=True,
                                       num_samples=pool_size(args.batch_size * world)))

        def one_step():
            batch = next(it)
            if batch[0].shape[0] != args.batch_size:
                raise RuntimeError(f"rank {rank} drew {batch[0].shape[0]} clips instead of {args.batch_size}: not a weak-scaling step")
            # pipelined: the next call is another step (the decoder layers' Adam slice runs under its encoder forward); the
            # timed region ends with a device-wide synchronize, which covers the last one
            return train.distributed_train_step(strategy, model, batch, opt, pipelined=True)
        # the same step through a launch plan (one replica: recorded once, then replayed from one C call; train.planned_step)
        plan_kind = "whisper"
        pstep = train.planned_step(strategy, model, opt, plan_kind, pipelined=True)
        cur = [pstep]

        def timed_step():
            batch = next(it)
            if batch[0].shape[0] != args.batch_size:
                raise RuntimeError(f"rank {rank} drew {batch[0].shape[0]} clips instead of {args.batch_size}: not a weak-scaling step")
            return cur[0](*batch)
        c = model.config
        metric = f"audio-seconds/sec/node (Whisper-{args.model_type}, 30 s clips)"
        workload = (f"whisper-{args.model_type}-ref (reference '{args.model_type}': {c.d_model}/{c.encoder_attention_heads}h/{c.d_ff}, "
                    f"{c.encoder_layers}+{c.decoder_layers} layers) train step, per-GPU batch {args.batch_size}, 30 s clips [80x3000], S=100")
        gf_sample = GF_PER_SAMPLE_BY_SIZE.get(args.model_type)
        tag = "whisper" if args.model_type == "small" else f"whisper_{args.model_type}"
    else:
        from tethys_speech_amd import wav2vec2
        from tethys_speech_amd.data import W2VDummyDataset
        clip = 2.0
        size = "base" if args.model_type == "small" else args.model_type
        model = wav2vec2.create_full_model("pretraining", size, device=dev, precision=args.precision, seed=1234)
        strategy.broadcast_parameters(model.arena.p)
        model.refresh_shadows()
        if args.dropout == "reference":
            c = model.config
            model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=1234 * 1000003 + rank, act_p=c.activation_dropout)
        opt = optim.Adam(learning_rate=3e-5, epsilon=1e-8)
        it = iter(W2VDummyDataset(args.batch_size, device=dev, rank=rank, world=world, seed=1234,
                                  num_samples=pool_size(args.batch_size * world)))
        rng = np.random.default_rng(1235)
        # the reference draws its negative indices on the device inside the step; here they are a step input, drawn with
        # its recipe: a ring of pre-drawn index sets already resident in HBM (like the audio batches)
        negs = [torch.from_numpy(wav2vec2.sample_negative_indices(rng, args.batch_size, 100, 100)).to(dev) for _ in range(64)]
        ctr = [0]

        def one_step():
            ctr[0] += 1
            audio = next(it)
            if audio.shape[0] != args.batch_size:
                raise RuntimeError(f"rank {rank} drew {audio.shape[0]} clips instead of {args.batch_size}: not a weak-scaling step")
            # (pipelined: see the Whisper step above)
            return train.wav2vec2_train_step(strategy, model, audio, negs[ctr[0] % len(negs)], opt, pipelined=True)
        plan_kind = "wav2vec2"
        pstep = train.planned_step(strategy, model, opt, plan_kind, pipelined=True)
        cur = [pstep]

        def timed_step():
            ctr[0] += 1
            audio = next(it)
            if audio.shape[0] != args.batch_size:
                raise RuntimeError(f"rank {rank} drew {audio.shape[0]} clips instead of {args.batch_size}: not a weak-scaling step")
            return cur[0](audio, negs[ctr[0] % len(negs)])
        metric = f"audio-seconds/sec/node (Wav2Vec2-{size} pretrain step, 2 s clips)"
        workload = f"wav2vec2-{size} pre-training step (V:), per-GPU batch {args.batch_size}, 2 s clips [32000]"
        gf_sample = W2V_GF_PER_SAMPLE.get(size)
        if args.workload == "whisper_single":
            # speech_jobs/whisper_single.py (S:): the same model, 5 s clips, roll-based negatives, no clipping, Adam eps 1e-7
            if world != 1:
                raise SystemExit("whisper_single is the single-device job (S:1303-1324)")
            clip = 5.0
            opt = optim.Adam(learning_rate=3e-5)
            it = iter(W2VDummyDataset(args.batch_size, length=80000, device=dev, seed=1234))
            model._prepare(args.batch_size, 80000)
            negs_t = [torch.from_numpy(wav2vec2.sample_negative_indices_roll(rng, model.T, model.config.num_negatives)).to(dev)
                      for _ in range(16)]

            def one_step():  # noqa: F811
                ctr[0] += 1
                return train.single_train_step(model, next(it), negs_t[ctr[0] % len(negs_t)], opt)
            pstep = train.planned_step(strategy, model, opt, "single")

            def timed_step():  # noqa: F811
                ctr[0] += 1
                return pstep(next(it), negs_t[ctr[0] % len(negs_t)])
            metric = "audio-seconds/sec/node (whisper_single.py = Wav2Vec2-base single-device step, 5 s clips)"
            workload = f"speech_jobs/whisper_single.py step (S:): wav2vec2-base, batch {args.batch_size}, 5 s clips [80000]"
            gf_sample = None
        tag = f"wav2vec2_{size}"

    log(f"model ready ({model.arena.n_params} params), warming up {args.warmup} steps")
    if args.no_plan:
        timed_step, pstep = one_step, one_step
        pstep.planned = None
    dt, last_loss, host_ms = timed_region(timed_step, args, strategy, dev, world)
    plan_note = None
    if getattr(pstep, "planned", None) is not None:
        pl = [v["plan"] for v in pstep.planned._by_sig.values() if v.get("plan") is not None]
        plan_note = {"replayed_steps": pstep.planned.replays, "launches": [p_.launches for p_ in pl], "nodes": [p_.nodes for p_ in pl],
                     "host_callbacks": [p_.callbacks for p_ in pl]}
        log(f"launch plan: {plan_note}")
    log(f"timed {args.steps} steps: {dt / args.steps * 1e3:.2f} ms/step, loss {last_loss:.4f}")

    multi = None
    if world > 1:
        # the same ranks without the exchange, issued the same way as the timed steps: a launch plan of its own (the recorded
        # launch sequence differs), or the eager step with --no-plan
        def without_exchange(run):
            if args.no_plan:
                return run(one_step)
            keep = cur[0]
            cur[0] = train.planned_step(strategy, model, opt, plan_kind, pipelined=True)
            try:
                return run(timed_step)
            finally:
                cur[0] = keep
        multi = exchange_diagnostics(without_exchange, args, strategy, dev, world, dt / args.steps * 1e3, host_ms)

    roof = classes = None
    if not args.no_roofline:
        roof, classes = measure_roofline(model, one_step, min(args.steps, 3), args.precision, tag)

    if rank == 0:
        gb = args.batch_size * world
        out = {
            "metric": metric, "value": throughput(clip, args.batch_size, world, args.steps, dt), "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": workload, "global_batch": gb, "parallelism": f"dp{world}", "last_loss": last_loss,
                       "dropout": drop_note, "grad_exchange_dtype": args.grad_dtype if world > 1 else None,
                       "exchange": args.exchange if world > 1 else None,
                       "host_enqueue_ms_per_step": host_ms,
                       "launch_plan": plan_note,
                       "step_tflops": gf_sample * gb * args.steps / dt / 1e3 if gf_sample else None},
        }
        if multi is not None:
            out["config"].update(multi)
        if roof is not None:
            out["roofline"] = roof
            out["roofline_classes"] = classes
        if world == 1 and not args.no_cpu_baseline:
            log("timing the restated reference CPU path (oracle) on the host cores")
            del model
            torch.cuda.empty_cache()
            if args.workload == "whisper":
                out["cpu_baseline"] = cpu_baseline_whisper(args.model_type, args.batch_size, dev, args.cpu_budget)
            else:
                out["cpu_baseline"] = cpu_baseline_w2v(size, args.batch_size, dev, args.cpu_budget,
                                                       single=args.workload == "whisper_single")
        # ONE write: the line must not be cut in two by another rank's output on the shared pipe
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    if world > 1:
        strategy.barrier()  # the other ranks stay silent until the line is out: library teardown messages come after it
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
