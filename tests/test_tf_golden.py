"""Dormant pin against the reference's own TensorFlow arithmetic (last row).  ``tools/export_tf_golden.py``,
run ONCE by someone with the reference's environment (NGC TF 22.12), writes ``tests/golden/tf/whisper_fixture.npz``:
initial weights by variable path, the losses of N steps of the reference's own ``distributed_train_step`` (dropout 0) and
three gradients of the first batch.  Until that file exists these tests SKIP and the oracle stays "parity unpinned"
(DESIGN (c)); once it is committed they hold
  * the oracle (fp64 on the CPU) - ``-m "not gpu"`` - and
  * the HIP fp32 path through ``arena.load_ref`` - ``-m gpu`` -
to the TensorFlow numbers: loss per step <= 1e-3 absolute (BASELINE.json's bound), gradients <= 1e-4 (normalised by
max |g|).  The exporter's own pool recipe is checked too (hash of the rebuilt pool), on CPU, fixture or not."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "tf", "whisper_fixture.npz")


def _exporter():
    spec = importlib.util.spec_from_file_location("export_tf_golden", os.path.join(ROOT, "tools", "export_tf_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # (TensorFlow is imported inside main() only)
    return mod


def _load():
    if not os.path.exists(FIXTURE):
        pytest.skip("tests/golden/tf/whisper_fixture.npz not exported yet (needs a TensorFlow box: tools/export_tf_golden.py)")
    z = np.load(FIXTURE, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    weights = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w:")}
    grads = {k[2:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith("g:")}
    return meta, weights, grads, [float(x) for x in z["losses"]], float(z["loss0"])


def _pool(meta):
    feats, labels = O.create_dummy_pool(seed=meta["seed"], n_mels=meta["config"]["n_mels"], seq_len=meta["seq_len"],
                                        max_target_length=meta["max_target_length"])
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(feats).tobytes())
    h.update(np.ascontiguousarray(labels).tobytes())
    assert h.hexdigest() == meta["pool_sha256"], "the exporter drew a different pool than the oracle rebuilds"
    return feats, labels


def test_exporter_pool_recipe_and_variable_walk_match_the_oracle():
    """Runs without the fixture: the NumPy-only restatements inside the exporter (the pool of W:784-815, the variable paths)
    are the oracle's, so a fixture exported tomorrow lines up with ``load_ref`` and the rebuilt batches."""
    ex = _exporter()
    f1, l1 = ex.dummy_pool(7, 16, 48, 12, num_samples=6)
    f2, l2 = O.create_dummy_pool(seed=7, n_mels=16, seq_len=48, max_target_length=12, num_samples=6)
    assert np.array_equal(f1, f2) and np.array_equal(l1, l2)
    f1, l1 = ex.dummy_pool(1234, 80, 300, 100, num_samples=5)
    f2, l2 = O.create_dummy_pool(seed=1234, n_mels=80, seq_len=300, max_target_length=100, num_samples=5)
    assert np.array_equal(f1, f2) and np.array_equal(l1, l2)
    cfg = O.make_config("small", d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=512,
                        encoder_layers=2, decoder_layers=2, decoder_start_token_id=511)
    assert ex.variable_paths(cfg) == list(O.param_shapes(cfg).keys())
    b1, b2 = ex.batches(f1, l1, 2), O.batches(f2, l2, 2)
    for _ in range(4):
        (fa, la), (fb, lb) = next(b1), next(b2)
        assert np.array_equal(fa, fb) and np.array_equal(la, lb)


def test_oracle_matches_tensorflow_fixture():
    meta, weights, grads, tf_losses, tf_loss0 = _load()
    feats, labels = _pool(meta)
    cfg = O.make_config("small", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, **meta["config"])
    params = {k: v.double() for k, v in weights.items()}
    it = O.batches(feats, labels, meta["batch"])
    f0, l0 = next(it)
    loss0, g0 = O.loss_and_grads(params, torch.from_numpy(np.ascontiguousarray(f0)), torch.from_numpy(np.ascontiguousarray(l0)), cfg)
    assert abs(float(loss0) - tf_loss0) <= 1e-4, (float(loss0), tf_loss0)
    for k, g in grads.items():
        err = float((g0[k].double() - g).abs().max() / g.abs().max())
        assert err <= 1e-4, (k, err)
    losses, _ = O.train_steps(cfg, params, feats, labels, meta["batch"], meta["steps"], lr=meta["lr"])
    err = max(abs(a - b) for a, b in zip(losses, tf_losses))
    assert err <= 1e-3, (err, losses, tf_losses)


@pytest.mark.gpu
def test_hip_fp32_path_matches_tensorflow_fixture(dev):
    meta, weights, grads, tf_losses, tf_loss0 = _load()
    feats, labels = _pool(meta)
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, optim, train, whisper
    model = whisper.create_whisper_model("small", device=dev, precision="fp32", **meta["config"])
    model.arena.load_ref(weights)
    model.refresh_shadows()
    it = O.batches(feats, labels, meta["batch"])
    f0, l0 = next(it)
    loss0 = model.forward_backward(torch.from_numpy(np.ascontiguousarray(f0)).to(dev), torch.from_numpy(np.ascontiguousarray(l0)).to(dev))
    assert abs(float(loss0.item()) - tf_loss0) <= 1e-4
    got = model.arena.ref_views(model.arena.g)
    for k, g in grads.items():
        err = float((got[k].double().cpu() - g).abs().max() / g.abs().max())
        assert err <= 1e-4, (k, err)
    model.arena.g.zero_()
    model.arena.g_clean = True
    opt = optim.Adam(learning_rate=meta["lr"])
    strat = dist.DataParallelStrategy(0, 1)
    it = O.batches(feats, labels, meta["batch"])
    losses = []
    for _ in range(meta["steps"]):
        f, l = next(it)
        losses.append(float(train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                                        torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt).item()))
    err = max(abs(a - b) for a, b in zip(losses, tf_losses))
    assert err <= 1e-3, (err, losses, tf_losses)
