"""SURVEY 8f row 3 on the GPU, through the job shim: ``speech_jobs/whisper_dist_tensorsize.py``'s ``main()`` runs the real
training job (tiny dimensions) with the tensor-size / skewness report beside it, as the reference drives its profiler
around the real step (speech_jobs/whisper_dist_tensorsize.py:1584-1613).  Checked: the seven report files, one row per
logged step, and the "Tiresias tensorsize" (mean step total after min(3, n // 4) warm-up steps, :207-222) against a sum
computed here from first principles (closed forms of the logged shapes, not tensorsize.py's list)."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVER = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
            encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=64, decoder_start_token_id=150, max_target_positions=32)
MB = 1024 * 1024


def _step_activation_elements(B, T, S, d, ff, H, Le, Ld):
    """Elements one step logs (reference names at :595-777): positional encodings in/out; per attention call the
    hidden-state input, k, v, q, scores, probabilities, raw and merged output (+ the [1,1,S,S] mask of the decoder's
    self-attention); per feed-forward its input, fc1 output, activation output, fc2 output and final output."""
    enc = 2 * B * T * d + Le * ((6 * B * T * d + 2 * B * H * T * T) + B * T * (3 * d + 2 * ff))
    dec_self = 6 * B * S * d + 2 * B * H * S * S + S * S
    dec_cross = 4 * B * S * d + 2 * B * T * d + 2 * B * H * S * T
    dec = 2 * B * S * d + Ld * (dec_self + dec_cross + B * S * (3 * d + 2 * ff))
    return enc + dec


def test_whisper_dist_tensorsize_main_writes_the_report(dev, tmp_path, monkeypatch):
    ws, res = str(tmp_path / "workspace"), str(tmp_path / "result")
    os.makedirs(ws)
    os.makedirs(os.path.join(res, "jobname"))
    with open(os.path.join(ws, "model.txt"), "w") as f:
        f.write("jobname\n")
    for k in ("TF_CONFIG", "RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setenv("TETHYS_WORKSPACE", ws)
    monkeypatch.setenv("TETHYS_RESULT", res)
    sys.path.insert(0, os.path.join(ROOT, "speech_jobs"))
    import whisper_dist_tensorsize as job
    B, NB, seq_len, S = 2, 9, 96, 12
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = job.main(["--batch_size", str(B), "--num_batches", str(NB)], model_overrides=OVER,
                      train_kw=dict(seq_len=seq_len, max_target_length=S))
    assert rc == 0
    out = buf.getvalue()
    logs = os.path.join(ws, "tensor_logs")
    for f in ("tensor_sizes.txt", "summary.txt", "tiresias_tensorsize.txt", "memory_usage.txt", "final_summary.json",
              "tiresias_result.json", "legacy_skewness_result.txt"):
        assert os.path.exists(os.path.join(logs, f)), f
    # the job really trained: NB step lines, a JCT file
    assert sum(1 for l in out.split("\n") if l.startswith("Step ")) == NB
    assert os.path.exists(os.path.join(res, "jobname", "None_None_jct.txt"))

    # independent totals.  Parameters: counted from the variable shapes of W:305-545 by hand
    d, ff, H, V, mels = 128, 256, 2, 160, 16
    ln = 2 * d
    mha = 4 * (d * d + d)
    ffn = d * ff + ff + ff * d + d
    n_params = ((3 * mels * d + d) + (3 * d * d + d) + (mha + ffn + 2 * ln) + ln          # conv1, conv2, 1 encoder layer, encoder LN
                + V * d + (2 * mha + ffn + 3 * ln) + ln + d * V)                           # embedding, 1 decoder layer, decoder LN, lm_head
    T = seq_len // 2                                                                       # conv2: stride 2, "same"
    step_bytes = 4 * (_step_activation_elements(B, T, S, d, ff, H, 1, 1) + n_params)       # activations + every gradient
    totals = [n_params * 4 / MB] + [step_bytes / MB] * NB                                  # the parameter pass, then NB steps
    warm = min(3, len(totals) // 4)
    tiresias = float(np.mean(totals[warm:]))

    rows = open(os.path.join(logs, "tiresias_tensorsize.txt")).read().splitlines()
    assert rows[0] == "step,tensorsize_mb" and len(rows) == 1 + 1 + NB
    got_steps = [float(r.split(",")[1]) for r in rows[1:]]
    assert np.allclose(got_steps, totals, rtol=0, atol=1e-4), (got_steps[:3], totals[:3])
    fin = json.load(open(os.path.join(logs, "final_summary.json")))
    tir = json.load(open(os.path.join(logs, "tiresias_result.json")))
    assert abs(fin["tiresias_tensorsize_mb"] - tiresias) <= 1e-6 * tiresias, (fin["tiresias_tensorsize_mb"], tiresias)
    assert tir["tensorsize_mb"] == fin["tiresias_tensorsize_mb"] and tir["total_steps"] == NB + 1
    assert tir["measurement_method"] == "Tiresias_style"
    # the skewness in the files is scipy's over the logged (non-empty) sizes
    from scipy import stats
    sizes = [int(l.split(",")[3]) / MB for l in open(os.path.join(logs, "tensor_sizes.txt")).read().splitlines()[1:]]
    assert abs(fin["model_skewness"] - stats.skew([s for s in sizes if s > 0])) < 1e-9
    # memory_usage.txt carries a live device figure for every step (the job ran on the GPU)
    mem = [l.split(",") for l in open(os.path.join(logs, "memory_usage.txt")).read().splitlines()[1:]]
    assert len(mem) == NB and all(float(m[1]) > 0 for m in mem)
    assert f"Tiresias tensorsize: {fin['tiresias_tensorsize_mb']:.2f} MB" in out
