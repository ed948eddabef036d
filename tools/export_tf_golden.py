#!/usr/bin/env python3
"""One-time exporter of a TensorFlow-side golden fixture (last row): the only route from "parity unpinned" to a
pin against the reference's own arithmetic.  NOT part of the build, the tests or the bench: nothing here runs in the
build container or on the GPU box (TensorFlow exists in neither).  A person with the reference's environment
(nvcr.io/nvidia/tensorflow:22.12-tf2-py3, Dockerfile:1 of the reference) runs it ONCE:

    python tools/export_tf_golden.py --reference /path/to/tethys-speech --out tests/golden/tf

and commits the resulting ``tests/golden/tf/whisper_fixture.npz`` (~6 MB).  ``tests/test_tf_golden.py`` then stops
skipping: it loads the initial weights through ``arena.load_ref`` / the oracle's parameter dict, replays the same batches
and holds the oracle (fp64, CPU) and the HIP fp32 path to the TensorFlow losses and gradients.

What it does, with the reference's OWN code (imported from ``--reference`` by path; nothing is copied):
  * ``WhisperConfig`` (W:10-45) shrunk to a fixture-sized model (--d_model 128, 2 + 2 layers, vocab 512: the weights must
    travel inside the fixture, 148 M parameters cannot); every ``Dropout`` layer's rate set to 0; NumPy / TF seeds fixed;
  * ``WhisperForConditionalGeneration(config)`` built by one call, its variables exported by ATTRIBUTE PATH
    (``encoder.layers.0.self_attn.q_proj.kernel`` ... - the reference names no layers, so Keras auto-names are useless);
  * the synthetic pool drawn with ``numpy.random.default_rng(seed)`` by the recipe of W:784-815 (restated below with NumPy
    only, because the NGC image has no torch; its SHA-256 is stored so the test can prove it rebuilt the same pool);
  * three gradients of the first batch (``tape.gradient`` as in W:829-832) BEFORE any update;
  * ``--steps`` steps of the reference's own ``distributed_train_step`` (W:819-848) under the default strategy with
    ``tf.keras.optimizers.Adam(1e-4)`` (W:901), batches taken as ``dataset.batch(B).repeat()`` does (W:812-815).
The headline configuration itself (small-ref, B = 8) is not exported: its 600 MB of initial weights cannot travel in a
fixture, and the arithmetic pinned here is the same code at a smaller width.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

GRAD_KEYS = ("lm_head.kernel", "encoder.conv1.kernel", "decoder.layers.0.encoder_attn.v_proj.kernel")


def dummy_pool(seed, n_mels, seq_len, max_target_length, num_samples=50):
    """W:784-815 with a seeded generator: features randn [N, n_mels, seq_len] fp32; labels [N, L] int32 with [0] = BOS 1,
    [1:len-1] = randint(3, 100), [len-1] = EOS 2, rest 0 (pad); len = randint(50, 90) (clipped to L).  Line for line the
    draw order of oracle/whisper_oracle.py:create_dummy_pool - the fixture stores a hash and the test checks it."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((num_samples, n_mels, seq_len)).astype(np.float32)
    labels = np.zeros((num_samples, max_target_length), dtype=np.int32)
    hi = min(90, max_target_length)
    lo = min(50, hi - 1)
    lengths = rng.integers(lo, hi, size=num_samples)
    for i in range(num_samples):
        labels[i, 0] = 1
        n = int(lengths[i])
        labels[i, 1:n - 1] = rng.integers(3, 100, size=n - 2)
        labels[i, n - 1] = 2
    return feats, labels


def pool_digest(feats, labels):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(feats).tobytes())
    h.update(np.ascontiguousarray(labels).tobytes())
    return h.hexdigest()


def batches(feats, labels, batch):
    """dataset.batch(batch).repeat() (W:812-815): the short last batch of a pass is kept."""
    n = feats.shape[0]
    while True:
        for s in range(0, n, batch):
            yield feats[s:s + batch], labels[s:s + batch]


def resolve(root, path):
    obj = root
    for part in path.split("."):
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj


def variable_paths(cfg):
    """Attribute paths of every trainable variable, in forward order (the key set of arena.load_ref / oracle.param_shapes)."""
    out = []
    for c in ("conv1", "conv2"):
        out += [f"encoder.{c}.kernel", f"encoder.{c}.bias"]

    def mha(p):
        return [f"{p}.{w}.{t}" for w in ("q_proj", "k_proj", "v_proj", "out_proj") for t in ("kernel", "bias")]

    def ln(p):
        return [f"{p}.gamma", f"{p}.beta"]

    def ffn(p):
        return [f"{p}.fc1.kernel", f"{p}.fc1.bias", f"{p}.fc2.kernel", f"{p}.fc2.bias"]
    for i in range(cfg.encoder_layers):
        p = f"encoder.layers.{i}"
        out += mha(p + ".self_attn") + ln(p + ".self_attn_layer_norm") + ffn(p + ".feed_forward") + ln(p + ".final_layer_norm")
    out += ln("encoder.layer_norm") + ["decoder.embed_tokens.embeddings"]
    for i in range(cfg.decoder_layers):
        p = f"decoder.layers.{i}"
        out += (mha(p + ".self_attn") + ln(p + ".self_attn_layer_norm") + mha(p + ".encoder_attn") +
                ln(p + ".encoder_attn_layer_norm") + ffn(p + ".feed_forward") + ln(p + ".final_layer_norm"))
    out += ln("decoder.layer_norm") + ["lm_head.kernel"]
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="checkout of hyunnnchoi/tethys-speech (read only; imported by path)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tf"))
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--d_model", type=int, default=128)
    ap.add_argument("--heads", type=int, default=2)
    ap.add_argument("--d_ff", type=int, default=256)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--vocab", type=int, default=512)
    ap.add_argument("--seq_len", type=int, default=3000)
    ap.add_argument("--max_target_length", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()

    import tensorflow as tf  # the reference's dependency; absent from the build container on purpose
    src = os.path.join(a.reference, "speech_jobs", "whisper_dist.py")
    spec = importlib.util.spec_from_file_location("tethys_ref_whisper_dist", src)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)  # (everything with side effects in that file sits under __main__)

    np.random.seed(a.seed)
    tf.random.set_seed(a.seed)
    cfg = ref.WhisperConfig()
    cfg.d_model, cfg.d_ff = a.d_model, a.d_ff
    cfg.encoder_layers = cfg.decoder_layers = a.layers
    cfg.encoder_attention_heads = cfg.decoder_attention_heads = a.heads
    cfg.vocab_size = a.vocab
    cfg.decoder_start_token_id = a.vocab - 1       # W:44's 50257 must stay inside the (shrunk) vocabulary
    cfg.dropout = cfg.attention_dropout = cfg.activation_dropout = 0.0
    model = ref.WhisperForConditionalGeneration(cfg)
    for layer in model.submodules:                  # belt and braces: any Dropout built from another rate
        if isinstance(layer, tf.keras.layers.Dropout):
            layer.rate = 0.0

    feats, labels = dummy_pool(a.seed, cfg.n_mels, a.seq_len, a.max_target_length)
    it = batches(feats, labels, a.batch)
    f0, l0 = next(it)
    model(tf.constant(f0), labels=tf.constant(l0), training=True)  # builds the variables
    paths = variable_paths(cfg)
    weights = {}
    for p in paths:
        root = model if p.startswith("lm_head") else model.model
        weights[p] = resolve(root, p).numpy().astype(np.float32)
    n_model = int(sum(int(np.prod(v.shape)) for v in model.trainable_variables))
    n_export = int(sum(v.size for v in weights.values()))
    if n_model != n_export:
        raise SystemExit(f"variable walk is incomplete: model has {n_model} trainable elements, exported {n_export}")

    # gradients of the first batch before any update (W:829-832)
    with tf.GradientTape() as tape:
        loss0 = model(tf.constant(f0), labels=tf.constant(l0), training=True)["loss"]
    by_id = {id(v): g for v, g in zip(model.trainable_variables, tape.gradient(loss0, model.trainable_variables))}
    grads = {}
    for k in GRAD_KEYS:
        var = resolve(model if k.startswith("lm_head") else model.model, k)
        g = by_id[id(var)]
        grads[k] = (tf.convert_to_tensor(g) if isinstance(g, tf.IndexedSlices) else g).numpy().astype(np.float32)

    # the reference's own step function, default (single-replica) strategy
    strategy = tf.distribute.get_strategy()
    optimizer = tf.keras.optimizers.Adam(learning_rate=a.lr)
    losses, sizes = [], []
    it = batches(feats, labels, a.batch)
    for _ in range(a.steps):
        f, l = next(it)
        sizes.append(int(f.shape[0]))
        loss = ref.distributed_train_step(strategy, model, (tf.constant(f), tf.constant(l)), optimizer)
        losses.append(float(loss.numpy()))
        print(f"step {len(losses) - 1}: loss {losses[-1]:.6f}", flush=True)

    os.makedirs(a.out, exist_ok=True)
    meta = {"tf_version": tf.__version__, "seed": a.seed, "lr": a.lr, "batch": a.batch, "steps": a.steps, "batch_sizes": sizes,
            "seq_len": a.seq_len, "max_target_length": a.max_target_length, "pool_sha256": pool_digest(feats, labels),
            "config": {"d_model": cfg.d_model, "encoder_attention_heads": cfg.encoder_attention_heads,
                       "decoder_attention_heads": cfg.decoder_attention_heads, "d_ff": cfg.d_ff,
                       "encoder_layers": cfg.encoder_layers, "decoder_layers": cfg.decoder_layers, "n_mels": cfg.n_mels,
                       "n_ctx": cfg.n_ctx, "vocab_size": cfg.vocab_size, "max_target_positions": cfg.max_target_positions,
                       "decoder_start_token_id": cfg.decoder_start_token_id},
            "grad_keys": list(GRAD_KEYS), "source": "speech_jobs/whisper_dist.py (distributed_train_step, W:819-848)"}
    path = os.path.join(a.out, "whisper_fixture.npz")
    np.savez_compressed(path, meta=json.dumps(meta), losses=np.asarray(losses, np.float64), loss0=np.float64(loss0.numpy()),
                        **{"w:" + k: v for k, v in weights.items()}, **{"g:" + k: v for k, v in grads.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1e6:.1f} MB): {n_export} parameters, {a.steps} losses, {len(grads)} gradients")


if __name__ == "__main__":
    sys.exit(main())
