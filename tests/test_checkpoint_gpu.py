"""Checkpoint save + restore (save: W:916-919,956 — the reference has no restore):
a restored (model, optimizer) continues on exactly the trajectory of the original."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_checkpoint_roundtrip_continues_training(dev, tmp_path):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, optim, train, whisper
    kw = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=32, decoder_start_token_id=150,
              max_target_positions=32)
    rng = np.random.default_rng(0)
    batches = [(torch.from_numpy(rng.standard_normal((2, 16, 48)).astype(np.float32)).to(dev),
                torch.from_numpy(rng.integers(0, 150, (2, 12)).astype(np.int32)).to(dev)) for _ in range(4)]
    strat = dist.DataParallelStrategy(0, 1)

    def run(model, opt, bs):
        return [float(train.distributed_train_step(strat, model, b, opt).item()) for b in bs]

    m1 = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=5, **kw)
    o1 = optim.Adam(1e-3)
    run(m1, o1, batches[:2])
    path = str(tmp_path / "ck.pt")
    train.save_checkpoint(m1, o1, path)
    path_bg = str(tmp_path / "ck_bg.pt")
    train.save_checkpoint(m1, o1, path_bg, background=True)  # snapshot now, written by a thread while training goes on
    ref = run(m1, o1, batches[2:])
    train.wait_for_checkpoints()
    a, b = torch.load(path, map_location="cpu"), torch.load(path_bg, map_location="cpu")
    assert all(torch.equal(a[k], b[k]) for k in ("p", "m", "v")) and a["iterations"] == b["iterations"] == 2

    m2 = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=99, **kw)  # different init
    o2 = optim.Adam(1e-3)
    train.load_checkpoint(m2, o2, path)
    assert o2.iterations == 2
    got = run(m2, o2, batches[2:])
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-6), (got, ref)

    # bf16 mode: the mirror is rebuilt from the restored master weights
    m3 = whisper.create_whisper_model("small", device=dev, precision="bf16", seed=99, **kw)
    o3 = optim.Adam(1e-3)
    train.load_checkpoint(m3, o3, path)
    w = m3.arena.p[:1024]
    assert torch.equal(m3.mirror[:1024], w.to(torch.bfloat16))

    # a checkpoint of another layout is refused
    kw2 = dict(kw, d_ff=128)
    m4 = whisper.create_whisper_model("small", device=dev, precision="fp32", **kw2)
    with pytest.raises(ValueError):
        train.load_checkpoint(m4, optim.Adam(1e-3), path)


def test_restore_in_place_resets_row_sparse_flags(dev, tmp_path):
    """Rollback into an (optimizer, model) pair that has already stepped: the row-activity flags of the row-sparse Adam
    (tmi_adam_step_rows) describe the OLD m / v.  Rows the pre-checkpoint batches touched have non-zero loaded m / v and
    must keep decaying although the flags of the rolled-back run never saw them: the restored trajectory has to equal the
    dense (row_sparse = False) one bit for bit."""
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, optim, train, whisper
    kw = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=32, decoder_start_token_id=150,
              max_target_positions=32)
    rng = np.random.default_rng(1)

    def batch(lo, hi):  # labels drawn from [lo, hi): different batches touch different embedding rows
        return (torch.from_numpy(rng.standard_normal((2, 16, 48)).astype(np.float32)).to(dev),
                torch.from_numpy(rng.integers(lo, hi, (2, 12)).astype(np.int32)).to(dev))
    early = [batch(3, 40), batch(3, 40)]       # rows 3..39 become active, then the checkpoint is taken
    late = [batch(60, 100), batch(60, 100), batch(60, 100)]
    strat = dist.DataParallelStrategy(0, 1)

    def run(model, opt, bs):
        return [float(train.distributed_train_step(strat, model, b, opt).item()) for b in bs]

    # dense reference trajectory
    md = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=5, **kw)
    od = optim.Adam(1e-3)
    od.row_sparse = False
    run(md, od, early)
    ref = run(md, od, late)

    # a second pair, row-sparse: steps on OTHER rows first (its flags mark rows 100..149 only), then rolls back in place
    ms = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=5, **kw)
    os_ = optim.Adam(1e-3)
    run(ms, os_, early)
    path = str(tmp_path / "ck.pt")
    train.save_checkpoint(ms, os_, path)
    m2 = whisper.create_whisper_model("small", device=dev, precision="fp32", seed=77, **kw)
    o2 = optim.Adam(1e-3)
    run(m2, o2, [batch(100, 150), batch(100, 150)])
    assert m2.arena.adam_row_flags  # the row-sparse path ran and keeps flags with the arena
    train.load_checkpoint(m2, o2, path)
    assert not m2.arena.adam_row_flags and m2.arena.adam_state_dirty
    got = run(m2, o2, late)
    assert np.allclose(got, ref, rtol=1e-6, atol=1e-6), (got, ref)

    def same_state(a, b):
        """Two runs of the same steps agree to the run-to-run noise of the step's fp32 atomics (last bits of the gradients);
        a row whose m / v stopped decaying or whose weights stopped moving (the stale-flag failure) is off by ~lr per step
        in p and by 1 - 0.999^k in v - orders of magnitude more."""
        off, rows, row_len = a.embedding_tables()[0]
        sl = slice(off, off + rows * row_len)
        for k, tol in (("p", 1e-5), ("m", 1e-6), ("v", 1e-9)):
            x, y = getattr(a.arena, k)[sl], getattr(b.arena, k)[sl]
            assert float((x - y).abs().max()) <= tol + 1e-4 * float(y.abs().max()), (k, float((x - y).abs().max()))
        assert float((a.arena.p - b.arena.p).abs().max()) <= 1e-5
    same_state(m2, md)
    # the rows only the pre-checkpoint batches touched kept moving after the restore (their gradient is zero now)
    off, rows, row_len = md.embedding_tables()[0]
    v_rows = m2.arena.v[off:off + rows * row_len].view(rows, row_len)
    assert float(v_rows[3:40].abs().sum()) > 0
    # switching the dense kernel on and off again must not resurrect stale flags either
    o2.row_sparse = False
    run(m2, o2, [late[0]]); run(md, od, [late[0]])
    o2.row_sparse = True
    run(m2, o2, [late[1]]); run(md, od, [late[1]])
    same_state(m2, md)
    # the flags are (re)created by a fill on torch's current stream while tmi_adam_step_rows reads and writes them on
    # the stream ops is pinned to (the second stream, for the early slice).  Invalidate them and step with either stream held
    # back by a long sleep kernel: a fill that lands after the row kernel would re-mark touched rows idle (or hand it
    # uninitialised flags) and those rows stop moving - same_state against the dense run shows it.
    for held in ("main", "side"):
        optim.Adam.invalidate_row_flags(m2)
        assert not m2.arena.adam_row_flags
        st = torch.cuda.current_stream() if held == "main" else m2._side
        if st is not None:
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(3e7))  # ~12 ms at 2.4 GHz: longer than the whole step
        run(m2, o2, [late[2]]); run(md, od, [late[2]])
        assert m2.arena.adam_row_flags
        same_state(m2, md)
