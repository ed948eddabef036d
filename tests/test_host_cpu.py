"""CPU-side checks: the C-ABI library loads and exports every declared symbol, host logic
(arena layout, strategy buckets, TF_CONFIG parsing, dataset batching), and the N>1
gradient all-reduce path with world_size-2 gloo."""
import ctypes
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "tethys_mi.h")).read()
    declared = set(re.findall(r"\b(tmi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"tmi_gemm_desc", "tmi_attn_desc"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = _lib.lib()  # loads, binds all symbols, checks the ABI version
    assert h.tmi_abi_version() == _lib.ABI_VERSION
    # struct layouts agree with the C compiler's
    src = '#include "%s"\n#include <stdio.h>\nint main(){printf("%%zu %%zu", sizeof(tmi_gemm_desc), sizeof(tmi_attn_desc));}' % os.path.join(ROOT, "include", "tethys_mi.h")
    import subprocess, tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", os.path.join(td, "s.c"), "-o", os.path.join(td, "s")])
        a, b = subprocess.check_output([os.path.join(td, "s")]).split()
    assert int(a) == ctypes.sizeof(_lib.GemmDesc) and int(b) == ctypes.sizeof(_lib.AttnDesc)


def test_bad_arguments_fail_loudly_without_a_gpu():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import _lib
    h = _lib.lib()
    assert h.tmi_gemm(None, None) == -1
    assert h.tmi_adam_step(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0.0, 1.0, None, 0, 0, None) == -1
    with pytest.raises(_lib.TmiError):
        _lib.check(h.tmi_layernorm_fwd(None, None, None, None, None, None, 0, 0, 1e-5, 0, None), "ln")


def test_arena_layout_and_reference_views():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import whisper
    from oracle import whisper_oracle as O
    cfg = whisper.make_config("small", d_model=64, encoder_attention_heads=1, decoder_attention_heads=1, d_ff=128,
                              vocab_size=131, encoder_layers=2, decoder_layers=2, n_mels=8)
    a = whisper.ParamArena(cfg, "cpu")
    ocfg = O.make_config("small", d_model=64, encoder_attention_heads=1, decoder_attention_heads=1, d_ff=128,
                         vocab_size=131, encoder_layers=2, decoder_layers=2, n_mels=8)
    shapes = O.param_shapes(ocfg)
    views = a.ref_views(a.p)
    assert set(views) == set(shapes)
    for k, s in shapes.items():
        assert tuple(views[k].shape) == tuple(s), k
    assert a.n_params == O.param_count(ocfg)
    assert all(o % 4 == 0 for o in a.offsets.values())
    # forward order: lm_head last, conv1 first -> backward fills the arena from the end
    assert a.names[0] == "encoder.conv1.kernel" and a.names[-1] == "lm_head.kernel"
    p = O.init_params(ocfg)
    a.load_ref(p)
    back = a.ref_views(a.p)
    for k in p:
        assert torch.equal(back[k], p[k])
    full = whisper.ParamArena(whisper.make_config("small"), "meta") if False else None  # (size only below)
    assert whisper.make_config("tiny").d_model == 384 and whisper.make_config("large").encoder_layers == 32


def test_strategy_buckets_and_env():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    s = D.DataParallelStrategy(0, 2, bucket_bytes=4 * 100, init=False)
    s.begin_gradients(torch.zeros(1000))
    with pytest.raises(RuntimeError):
        s.gradients_ready(100, 900)  # not adjacent to the end of the arena
    s = D.DataParallelStrategy(0, 1, bucket_bytes=4 * 100)
    b = s.buckets(250)
    assert b == [(150, 250), (50, 150), (0, 50)]
    assert D.task_from_env({}) == ("worker", 0, 0, 1)
    assert D.task_from_env({"RANK": "3", "WORLD_SIZE": "8"}) == ("worker", 3, 3, 8)
    tfc = '{"cluster":{"chief":["a:1"],"worker":["b:1","c:1"]},"task":{"type":"worker","index":1}}'
    assert D.task_from_env({"TF_CONFIG": tfc}) == ("worker", 1, 2, 3)
    tfc = '{"cluster":{"chief":["a:1"],"worker":["b:1"]},"task":{"type":"chief","index":0}}'
    assert D.task_from_env({"TF_CONFIG": tfc}) == ("chief", 0, 0, 2)


def test_dataset_batching_matches_reference_recipe():
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import data
    from oracle import whisper_oracle as O
    ds = data.create_dummy_dataset(8, n_mels=4, seq_len=16, max_target_length=100, device="cpu")
    f, l = O.create_dummy_pool(seed=1234, n_mels=4, seq_len=16)
    it = O.batches(f, l, 8)
    for _ in range(9):
        a, b = next(ds)
        fa, la = next(it)
        assert np.array_equal(a.numpy(), fa) and np.array_equal(b.numpy(), la)
    d0 = data.create_dummy_dataset(4, n_mels=4, seq_len=16, device="cpu", rank=0, world=2, drop_remainder=True)
    d1 = data.create_dummy_dataset(4, n_mels=4, seq_len=16, device="cpu", rank=1, world=2, drop_remainder=True)
    for step in range(8):
        a0, _ = next(d0)
        a1, _ = next(d1)
        s = (step % 6) * 8
        assert np.array_equal(a0.numpy(), f[s:s + 4]) and np.array_equal(a1.numpy(), f[s + 4:s + 8])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    from oracle import whisper_oracle as O
    torch.set_num_threads(2)
    strat = D.DataParallelStrategy(rank, world, backend="gloo", bucket_bytes=4 * 1000)
    cfg = O.make_config("small", d_model=32, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=64,
                        vocab_size=128, encoder_layers=1, decoder_layers=1, n_mels=8, n_ctx=12,
                        decoder_start_token_id=127, dropout=0.0, attention_dropout=0.0)
    f, l = O.create_dummy_pool(seed=2, n_mels=8, seq_len=24, max_target_length=6, num_samples=4)
    p = O.init_params(cfg, seed=100 + rank, dtype=torch.float64)  # deliberately different per rank
    names = sorted(p)
    flat = torch.cat([p[k].reshape(-1) for k in names])
    strat.broadcast_parameters(flat)  # C4
    off = 0
    for k in names:
        n = p[k].numel()
        p[k] = flat[off:off + n].view_as(p[k]).clone()
        off += n
    loss, g = O.loss_and_grads(p, torch.from_numpy(f[2 * rank:2 * rank + 2]), torch.from_numpy(l[2 * rank:2 * rank + 2]), cfg)
    gflat = torch.cat([g[k].reshape(-1) for k in names])
    # C1, overlapped form: ranges become final last-first, buckets fly while "backward" continues
    strat.begin_gradients(gflat)
    n = gflat.numel()
    strat.gradients_ready(2 * n // 3, n)
    strat.gradients_ready(n // 3, 2 * n // 3)
    launched = n - strat._pend_hi  # elements already handed to the exchange while "backward" was still running
    strat.all_reduce_gradients(gflat)  # sends the remaining head and waits for every bucket
    assert launched > 0
    g2 = torch.cat([g[k].reshape(-1) for k in names])
    strat.all_reduce_gradients(g2)     # plain bucketed form gives the same sums
    assert torch.equal(g2, gflat)
    tot = strat.reduce_sum(loss.reshape(1).clone())  # C2
    q.put((rank, gflat.numpy(), float(tot), flat.numpy()))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_single_process_sum():
    from oracle import whisper_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (r0, g0, l0, p0), (r1, g1, l1, p1) = res
    assert np.array_equal(g0, g1) and l0 == l1 and np.array_equal(p0, p1)
    cfg = O.make_config("small", d_model=32, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=64,
                        vocab_size=128, encoder_layers=1, decoder_layers=1, n_mels=8, n_ctx=12,
                        decoder_start_token_id=127, dropout=0.0, attention_dropout=0.0)
    f, l = O.create_dummy_pool(seed=2, n_mels=8, seq_len=24, max_target_length=6, num_samples=4)
    p = O.init_params(cfg, seed=100, dtype=torch.float64)  # rank 0's values were broadcast
    names = sorted(p)
    la, ga = O.loss_and_grads(p, torch.from_numpy(f[:2]), torch.from_numpy(l[:2]), cfg)
    lb, gb = O.loss_and_grads(p, torch.from_numpy(f[2:4]), torch.from_numpy(l[2:4]), cfg)
    ref = torch.cat([(ga[k] + gb[k]).reshape(-1) for k in names]).numpy()
    assert np.allclose(g0, ref, rtol=1e-12, atol=1e-15)
    assert l0 == pytest.approx(float(la) + float(lb), rel=1e-12)


def test_rendezvous_from_tf_config():
    """the rendezvous address comes from the cluster spec (rank 0's host:port) unless MASTER_* is set."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    tfc = '{"cluster":{"chief":["whisper-chief-0.ns.svc:2222"],"worker":["whisper-worker-0.ns.svc:2222"]},"task":{"type":"worker","index":0}}'
    assert D.rendezvous_from_env({"TF_CONFIG": tfc}) == ("whisper-chief-0.ns.svc", "2222")
    assert D.rendezvous_from_env({"TF_CONFIG": tfc, "MASTER_ADDR": "10.0.0.5"}) == ("10.0.0.5", "2222")
    assert D.rendezvous_from_env({"TF_CONFIG": tfc, "MASTER_ADDR": "10.0.0.5", "MASTER_PORT": "1"}) == ("10.0.0.5", "1")
    tfw = '{"cluster":{"worker":["w0:7000","w1:7000"]},"task":{"type":"worker","index":1}}'
    assert D.rendezvous_from_env({"TF_CONFIG": tfw}) == ("w0", "7000")
    assert D.rendezvous_from_env({}) == (None, None)


_TINY = dict(d_model=32, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=64, vocab_size=128,
             encoder_layers=2, decoder_layers=2, n_mels=8, n_ctx=12, decoder_start_token_id=127)


def _ragged_worker(rank, world, port, q):
    """Rank 0 has a real slice of the short final batch, rank 1 an empty one (train.py:22): both must
    issue the same bucket collectives in the same order.  The HIP model cannot run here, so rank 0's backward is
    played back (oracle gradients, reported in grad_ready_names() order); rank 1 runs the product's own empty-slice
    path (report_zero_gradients)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D, whisper
    from oracle import whisper_oracle as O
    torch.set_num_threads(2)
    strat = D.DataParallelStrategy(rank, world, backend="gloo", bucket_bytes=4 * 2000)
    model = whisper.create_whisper_model("small", device="cpu", precision="fp32", seed=5, **_TINY)
    a = model.arena
    launches = []
    orig = strat._launch

    def counting_launch():
        if strat._pend_hi > strat._pend_lo:
            launches.append((strat._pend_lo, strat._pend_hi))
        orig()
    strat._launch = counting_launch
    strat.begin_gradients(a.g)
    if rank == 0:
        cfg = O.make_config("small", dropout=0.0, attention_dropout=0.0, **_TINY)
        f, l = O.create_dummy_pool(seed=2, n_mels=8, seq_len=24, max_target_length=6, num_samples=1)
        p = {k: v.double() for k, v in a.ref_views(a.p).items()}
        _, g = O.loss_and_grads(p, torch.from_numpy(f), torch.from_numpy(l), cfg)
        for k, v in a.ref_views(a.g).items():
            v.copy_(g[k].float())
        want = a.g.clone()
        hi = a.numel
        for name in model.grad_ready_names():
            lo = a.offsets[name]
            if lo < hi:
                strat.gradients_ready(lo, hi)
                hi = lo
    else:
        want = None
        model.report_zero_gradients(strat.gradients_ready)
    strat.all_reduce_gradients(a.g)
    q.put((rank, launches, a.g.numpy().copy(), None if want is None else want.numpy()))
    dist.destroy_process_group()


def test_empty_slice_replica_issues_the_same_collectives():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, l0, g0, want), (_, l1, g1, _) = res
    assert l0 == l1 and len(l0) >= 3          # same buckets, same order, more than one of them
    assert l0[0][1] > l0[-1][1] and l0[-1][0] == 0
    assert np.array_equal(g0, g1) and np.array_equal(g0, want)  # zeros + rank 0's gradients


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist as D
    torch.set_num_threads(2)
    base = D.DataParallelStrategy(rank, world, backend="gloo", bucket_bytes=4 * 1000)
    gen = torch.Generator().manual_seed(100 + rank)
    n = 10_007  # not a multiple of the bucket size, the world size or the 8-element piece alignment
    g0 = torch.randn(n, generator=gen)
    out = {}
    for exchange in ("allreduce", "rs_ag", "mesh"):
        for gd in ("fp32", "bf16"):
            s = D.DataParallelStrategy(rank, world, bucket_bytes=4 * 1000, init=False, grad_dtype=gd, exchange=exchange)
            g = g0.clone()
            # overlapped form: three reports, then the rest
            s.begin_gradients(g)
            s.gradients_ready(7000, n)
            s.gradients_ready(3001, 7000)
            s.all_reduce_gradients(g)
            g2 = g0.clone()
            s.all_reduce_gradients(g2)  # plain bucketed form
            out[(exchange, gd)] = (g.numpy().copy(), g2.numpy().copy())
    q.put((rank, g0.numpy(), out))
    dist.destroy_process_group()


def test_exchange_forms_and_bf16_wire_two_rank_gloo():
    """reduce-scatter + all-gather and the all-links mesh form give the all-reduce's sums; bf16 on
    the wire stays within bf16 rounding of the fp32 exchange (2 addends: <= 2^-8 relative per element of the larger
    addend's magnitude)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, a, out0), (_, b, out1) = res
    want = a + b
    for key in out0:
        for got in (*out0[key], *out1[key]):
            if key[1] == "fp32":
                assert np.allclose(got, want, rtol=0, atol=1e-6), key
            else:
                bound = (np.abs(a) + np.abs(b)) * 2.0 ** -7 + 1e-6
                assert np.all(np.abs(got - want) <= bound), (key, float(np.abs(got - want).max()))
        assert np.array_equal(out0[key][0], out1[key][0]), key  # replicas end with identical gradients


def _bench_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import types
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import data, dist as D
    import bench
    strat = D.DataParallelStrategy(rank, world, backend="gloo")
    ds = iter(data.create_dummy_dataset(4, n_mels=4, seq_len=16, device="cpu", rank=rank, world=world, seed=1234,
                                        drop_remainder=True))
    seen = []

    def one_step():  # a stand-in step: rank 1 is the slow replica
        f, _ = next(ds)
        seen.append(f[:, 0, 0].clone())
        time.sleep(0.02 if rank == 0 else 0.05)
        return torch.tensor([float(rank)])
    args = types.SimpleNamespace(steps=5, warmup=2)
    dt, last, host_ms = bench.timed_region(one_step, args, strat, "cpu", world)
    nseen = len(seen)
    # the N > 1 diagnostics of the bench line (VERDICT r2 item 5 ii): every rank calls it, collectives inside
    diag = bench.exchange_diagnostics(lambda run: run(one_step), args, strat, "cpu", world, dt / args.steps * 1e3, host_ms)
    assert strat.exchange_off is False
    q.put((rank, dt, torch.stack(seen[:nseen]).numpy(), bench.throughput(30.0, 4, world, args.steps, dt), diag))
    dist.destroy_process_group()


def test_bench_timing_logic_two_rank_gloo():
    """bench.py's N > 1 contract on CPU: the timed region is the MAX over ranks (every rank reports
    the slow replica's time), replicas read disjoint slices of each global batch, value counts the global batch."""
    from oracle import whisper_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
    (_, dt0, seen0, v0, diag0), (_, dt1, seen1, v1, diag1) = res
    assert dt0 == dt1 and v0 == v1
    for dg in (diag0, diag1):
        assert dg["rccl_ranks"] == 2 and dg["backend"] == "gloo" and len(dg["host_enqueue_ms"]) == 2
        # the stand-in step has no exchange: with and without it the slow rank's 50 ms per step
        assert abs(dg["exposed_exchange_ms"]) < 15.0 and 45.0 < dg["ms_per_step_without_exchange"] < 70.0
    assert diag0["host_enqueue_ms"] == diag1["host_enqueue_ms"] and diag0["host_enqueue_ms"][1] > diag0["host_enqueue_ms"][0]
    assert dt0 >= 5 * 0.05 and dt0 < 5 * 0.05 + 0.5        # the slow rank's 5 timed steps, not the fast rank's
    assert v0 == pytest.approx(30.0 * 4 * 2 * 5 / dt0)
    f, _ = O.create_dummy_pool(seed=1234, n_mels=4, seq_len=16)
    for step in range(7):                                    # 2 warm-up + 5 timed global batches of 8
        s = (step % 6) * 8
        assert np.array_equal(seen0[step], f[s:s + 4, 0, 0]) and np.array_equal(seen1[step], f[s + 4:s + 8, 0, 0])


def test_stable_jobs_whisper_dist_is_the_speech_jobs_entry_point():
    """stable_jobs/whisper_dist.py is byte-identical to speech_jobs/whisper_dist.py in the reference: one main()."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("stable_whisper_dist", os.path.join(root, "stable_jobs", "whisper_dist.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from speech_jobs import whisper_dist
    assert mod.main is whisper_dist.main


def test_w2v_dataset_keeps_the_short_batch_across_replicas():
    """stable_jobs/wav2vec2_dist.py:1094-1111, 1226: batch(GLOBAL).repeat() over 50 clips without drop_remainder; replica r
    takes rows [r*B, (r+1)*B) of each global batch, short or empty on the last one."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd.data import W2VDummyDataset
    sizes = []
    for r in range(2):
        ds = W2VDummyDataset(4, length=16, device="cpu", rank=r, world=2, seed=1, drop_remainder=False)
        it = iter(ds)
        sizes.append([int(next(it).shape[0]) for _ in range(8)])
    assert sizes[0] == [4, 4, 4, 4, 4, 4, 2, 4] and sizes[1] == [4, 4, 4, 4, 4, 4, 0, 4]
    a = W2VDummyDataset(4, length=16, device="cpu", rank=0, world=2, seed=1, drop_remainder=False)
    b = W2VDummyDataset(4, length=16, device="cpu", rank=1, world=2, seed=1, drop_remainder=False)
    one = W2VDummyDataset(8, length=16, device="cpu", seed=1, drop_remainder=False)
    import torch
    assert torch.equal(torch.cat([next(iter(a)), next(iter(b))]), next(iter(one)))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bench_pool_gives_every_rank_its_full_batch(world):
    """bench.py --gpus N is a weak-scaling run: per-GPU batch 8 at every N.  The reference's 50-clip pool cannot feed a
    global batch of 64, so bench.pool_size grows it; every rank must draw 8 clips on every step, disjoint from its peers'."""
    import bench
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd.data import W2VDummyDataset, create_dummy_dataset
    n = bench.pool_size(8 * world)
    assert n >= 50 and (n == 50 or n >= 2 * 8 * world)
    its = [iter(create_dummy_dataset(8, n_mels=2, seq_len=4, max_target_length=6, device="cpu", rank=r, world=world, seed=1,
                                     drop_remainder=True, num_samples=n)) for r in range(world)]
    w2v = [iter(W2VDummyDataset(8, length=8, device="cpu", rank=r, world=world, seed=1, num_samples=n)) for r in range(world)]
    for _ in range(5):
        rows = [next(it)[0] for it in its]
        assert all(f.shape[0] == 8 for f in rows)
        flat = torch.cat(rows).reshape(8 * world, -1)
        assert len({tuple(x.tolist()) for x in flat}) == 8 * world  # distinct clips on every rank
        assert all(next(it).shape[0] == 8 for it in w2v)



def test_bench_self_launch_relays_rank0_line(tmp_path, capfd):
    """`python bench.py --gpus N` with no WORLD_SIZE must not lose the run.  bench.self_launch starts
    `python -m torch.distributed.run --nproc-per-node N <script> <argv>` as a CHILD process (never exec), relays its stdout
    and prints rank 0's JSON line as the last stdout line; the return code is the child's.  Here the ranks are a stand-in
    script (the bench's own ranks need a GPU: tests/test_bench_contract_gpu.py runs the real thing with gloo on one
    card): two gloo ranks all-reduce their rank + 1 and rank 0 prints the line."""
    import bench
    script = tmp_path / "ranks.py"
    script.write_text(
        "import json, os, sys, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(dist.get_rank() + 1)]); dist.all_reduce(t)\n"
        "print('noise from rank', dist.get_rank(), flush=True)\n"
        "if dist.get_rank() == 0:\n"
        "    print(json.dumps({'metric': 'm', 'value': t.item(), 'n_gpus': int(os.environ['WORLD_SIZE']), 'argv': sys.argv[1:]}), flush=True)\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(os.environ.get('RANKS_RC', '0')))\n")
    rc = bench.self_launch(2, script=str(script), argv=["--gpus", "2", "--steps", "3"])
    out = [l for l in capfd.readouterr().out.splitlines() if l.strip()]
    assert rc == 0
    d = json.loads(out[-1])  # the JSON line is the LAST line whatever the ranks printed after it
    assert d == {"metric": "m", "value": 3.0, "n_gpus": 2, "argv": ["--gpus", "2", "--steps", "3"]}
    assert "\n".join(out).count("noise from rank") == 2
    # the ranks share one pipe: another rank's (or a library's) output can land in the middle of rank 0's line.  The object is
    # found wherever it sits and what surrounds it is relayed (seen as a 1-in-10 failure of this test before the launcher
    # looked inside lines: 'noise from rank{"metric": ...}' + ' 1')
    cut = tmp_path / "cut.py"
    cut.write_text(
        "import json, os, sys, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "if dist.get_rank() == 0:\n"
        "    print('noise from rank' + json.dumps({'metric': 'm', 'value': 1.5, 'n_gpus': 2}) + ' 1', flush=True)\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    assert bench.self_launch(2, script=str(cut), argv=[]) == 0
    out = [l for l in capfd.readouterr().out.splitlines() if l.strip()]
    assert json.loads(out[-1]) == {"metric": "m", "value": 1.5, "n_gpus": 2}
    assert any(l.replace(" ", "") == "noisefromrank1" for l in out[:-1])
    os.environ["RANKS_RC"] = "3"
    try:
        assert bench.self_launch(2, script=str(script), argv=[]) != 0  # a failing rank fails the launcher
    finally:
        del os.environ["RANKS_RC"]
    capfd.readouterr()


def test_no_valu_read_inside_an_mfma_hazard_window():
    """VERDICT r4 item 6, static half: the disassembly of the built library has no VALU / memory instruction that reads an
    MFMA result fewer than the required wait states after it on a fall-through path (tools/check_mfma_hazards.py).  hipcc
    pads these for its own code but not inside an inline-asm statement - the round-4 attention bug (an asm v_max3_f32 on
    fresh score accumulators: 3 % of the outputs differed between identical launches).  The dynamic half is
    tests/test_kernels_gpu.py::test_kernels_are_bit_reproducible_under_load."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tethys-speech_amd", "libtethys_mi.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_mfma_hazards.py"), lib], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 candidate hazard(s) in" in r.stdout and " 0 code object" not in r.stdout
    # second rule (round 5): no SGPR is read or written while a scalar load into it may still be in flight - the dK/dV
    # kernel issues its mask loads by hand, and an asm output the compiler believed dead was handed to an address computation
    # (a fault on the one shape whose register allocation put a pointer there)
    assert "0 scalar-load destination(s) touched in flight" in r.stdout


def test_launch_plan_callback_nodes_replay_in_order_and_raise():
    """plan.host_call (the collectives of a job with replicas inside a recorded step): the function runs at once while the
    plan records, every replay calls it again at its place of the sequence, a ``WorkSlot`` is refilled by each replay, and
    an exception inside a callback comes out of ``replay`` (no GPU: the plan holds callback nodes only)."""
    from tethys_speech_amd import plan as P
    seen, slot = [], P.WorkSlot()

    class Work:
        def __init__(self, tag):
            self.tag = tag

        def wait(self):
            seen.append(("wait", self.tag))

    issued = [0]

    def issue():
        issued[0] += 1
        slot.work = Work(issued[0])
        seen.append(("issue", issued[0]))

    P.host_call(lambda: seen.append("outside"))   # no plan recording: just a call
    pl = P.LaunchPlan()
    with pl.recording():
        P.host_call(issue)
        slot.wait()
    assert seen == ["outside", ("issue", 1), ("wait", 1)] and pl.callbacks == 2 and pl.nodes == 2 and pl.launches == 0
    pl.replay()
    pl.replay()
    assert seen[3:] == [("issue", 2), ("wait", 2), ("issue", 3), ("wait", 3)]
    P.host_call(lambda: seen.append("after"))     # the recording is over: not a node
    assert pl.callbacks == 2

    bad = P.LaunchPlan()
    fail = [False]

    def maybe():
        if fail[0]:
            raise RuntimeError("collective failed")
    later = []
    with bad.recording():
        P.host_call(maybe)
        P.host_call(lambda: later.append(1))
    fail[0] = True
    with pytest.raises(RuntimeError, match="collective failed"):
        bad.replay()
    assert later == [1], "nodes after a failed callback must not issue anything"
    fail[0] = False
    bad.replay()
    assert later == [1, 1]
