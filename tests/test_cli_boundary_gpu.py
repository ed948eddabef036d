"""The process/CLI surface of the job shims: ``speech_jobs/whisper_dist.py``,
``wav2vec2_dist.py`` and ``whisper_single.py`` run through their ``main()`` with a TF_CONFIG naming one chief
and one worker (two processes sharing cuda:0, gloo carrying the buckets), TETHYS_WORKSPACE / TETHYS_RESULT in a
temp dir.  Asserted byte for byte against the reference's own print statements (W:951, 991-1000, 1012-1013,
1026, 1055-1056; V:1381-1408, 1424-1425, 1440, 1479-1481; S:1266-1274, 1286-1287, 1303, 1321-1322) and result
files (W:1016-1021: ``<type>_<index>_jct.txt`` = '%.2f' % jct; S:1292: ``single_jct.txt``)."""
import contextlib
import io
import json
import os
import re
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W_OVER = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
              encoder_layers=1, decoder_layers=1, n_mels=16, n_ctx=64, decoder_start_token_id=150, max_target_positions=32)
V_OVER = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
              conv_dim=(64, 64, 64), conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2), num_conv_pos_embeddings=8,
              num_conv_pos_embedding_groups=4, num_codevectors_per_group=16, codevector_dim=32,
              proj_codevector_dim=64, num_negatives=10)
STEP_RE = re.compile(r"^Step (\d+), Loss: -?\d+\.\d{4}, Time: \d\d:\d\d:\d\d \(경과: \d+\.\d\d초, 스텝 시간: \d+\.\d\d초\)$")
BANNER = ["", "========================", "network profile started!", "========================"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _job(which, ttype, tindex, port, ws, res, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "speech_jobs"))
    os.environ["TF_CONFIG"] = json.dumps({"cluster": {"chief": [f"127.0.0.1:{port}"], "worker": ["127.0.0.1:1"]},
                                          "task": {"type": ttype, "index": tindex}})
    os.environ.pop("MASTER_ADDR", None)
    os.environ.pop("MASTER_PORT", None)  # the rendezvous must come from the cluster spec
    os.environ.pop("RANK", None)
    os.environ.pop("WORLD_SIZE", None)
    os.environ["LOCAL_RANK"] = "0"
    os.environ["TETHYS_DIST_BACKEND"] = "gloo"  # two ranks on one device: RCCL cannot, gloo can
    os.environ["TETHYS_WORKSPACE"], os.environ["TETHYS_RESULT"] = ws, res
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        if which == "stable_w2v":
            import importlib.util
            spec = importlib.util.spec_from_file_location("stable_wav2vec2_dist", os.path.join(ROOT, "stable_jobs", "wav2vec2_dist.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            rc = mod.main(["--batch_size", "2", "--num_batches", "3"], model_overrides=V_OVER, train_kw=dict(clip_samples=800))
        elif which == "whisper":
            import whisper_dist
            rc = whisper_dist.main(["--batch_size", "2", "--num_batches", "3"], model_overrides=W_OVER,
                                   train_kw=dict(seq_len=96, max_target_length=12))
        else:
            import wav2vec2_dist
            rc = wav2vec2_dist.main(["--batch_size", "2", "--num_batches", "3", "--model_size", "base"],
                                    model_overrides=V_OVER, train_kw=dict(clip_samples=800))
    q.put((ttype, rc, buf.getvalue()))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def _run_pair(which, tmp_path):
    ws, res = str(tmp_path / "workspace"), str(tmp_path / "result")
    os.makedirs(ws)
    os.makedirs(os.path.join(res, "jobname"))
    with open(os.path.join(ws, "model.txt"), "w") as f:
        f.write("jobname\n")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job, args=(which, t, 0, port, ws, res, q)) for t in ("chief", "worker")]
    for p_ in procs:
        p_.start()
    out = dict((t, (rc, txt)) for t, rc, txt in [q.get(timeout=600) for _ in range(2)])
    for p_ in procs:
        p_.join(120)
    return ws, res, out


def _check_common(lines, head, n_steps):
    assert lines[:len(head)] == head, lines[:len(head) + 2]
    rest = lines[len(head):]
    assert rest[0] == "Epoch 1/1"
    steps = rest[1:1 + n_steps]
    for i, l in enumerate(steps):
        m = STEP_RE.match(l)
        assert m and int(m.group(1)) == i, l
    assert rest[1 + n_steps] == "Training completed."
    assert re.match(r"^jct: \d+\.\d+$", rest[2 + n_steps])
    return rest[3 + n_steps:], float(rest[2 + n_steps].split()[1]), [float(l.split("Loss: ")[1].split(",")[0]) for l in steps]


def test_whisper_dist_main_chief_and_worker(dev, tmp_path):
    ws, res, out = _run_pair("whisper", tmp_path)
    losses = {}
    for t in ("chief", "worker"):
        rc, txt = out[t]
        assert rc == 0
        lines = txt.split("\n")
        head = ["batch size per replica: 2, global batch size: 4", "num_batches: 3", "Whisper-small 분산 학습 시작..."] + BANNER
        tail, jct, losses[t] = _check_common(lines, head, 3)
        assert tail == [f"모델이 {os.path.join(ws, 'model_cache', 'whisper_small_model')}에 저장되었습니다.", ""]
        got = open(os.path.join(res, "jobname", f"{t}_0_jct.txt")).read()
        assert got == "%.2f" % jct
    assert losses["chief"] == losses["worker"]  # C2: both replicas print the reduced (summed) loss
    assert os.path.exists(os.path.join(ws, "model_cache", "whisper_small_model"))
    assert any(f.endswith(".pt") for f in os.listdir(os.path.join(ws, "checkpoints")))


def test_wav2vec2_dist_main_chief_and_worker(dev, tmp_path):
    ws, res, out = _run_pair("w2v", tmp_path)
    losses = {}
    for t in ("chief", "worker"):
        rc, txt = out[t]
        assert rc == 0
        lines = txt.split("\n")
        head = ["선택된 모델 크기: base", "batch size per replica: 2, global batch size: 4", "num_batches: 3",
                "Wav2Vec2 분산 학습 시작...", "선택된 모델 크기: base", "Base 모델: 약 95M 파라미터",
                "16GB V100 GPU에 최적화된 설정"] + BANNER
        tail, jct, losses[t] = _check_common(lines, head, 3)
        assert tail == [f"Base 모델이 {os.path.join(ws, 'model_cache', 'wav2vec2_base_model')}에 저장되었습니다.", ""]
        assert open(os.path.join(res, "jobname", f"{t}_0_jct.txt")).read() == "%.2f" % jct
    assert losses["chief"] == losses["worker"]


def test_whisper_single_main(dev, tmp_path, monkeypatch, capsys):
    """BASELINE config #1 read as the file is named (S:): single process, no TF_CONFIG, ``single_jct.txt``."""
    sys.path.insert(0, os.path.join(ROOT, "speech_jobs"))
    ws, res = str(tmp_path / "workspace"), str(tmp_path / "result")
    os.makedirs(ws)
    os.makedirs(os.path.join(res, "jobname"))
    with open(os.path.join(ws, "model.txt"), "w") as f:
        f.write("jobname\n")
    monkeypatch.setenv("TETHYS_WORKSPACE", ws)
    monkeypatch.setenv("TETHYS_RESULT", res)
    monkeypatch.delenv("TF_CONFIG", raising=False)
    import whisper_single
    rc = whisper_single.main(["--batch_size", "4", "--num_batches", "3"], model_overrides=V_OVER, clip_samples=2000)
    assert rc == 0
    lines = capsys.readouterr().out.split("\n")
    head = ["batch size: 4", "num_batches: 3", "Wav2Vec2 단일 GPU 학습 시작...", "", "========================",
            "GPU profile started!", "========================"]
    tail, jct, losses = _check_common(lines, head, 3)
    assert tail == [f"모델이 {os.path.join(ws, 'model_cache', 'wav2vec2_model')}에 저장되었습니다.", ""]
    assert open(os.path.join(res, "jobname", "single_jct.txt")).read() == "%.2f" % jct
    assert all(l == l for l in losses)


def test_stable_wav2vec2_dist_main_chief_and_worker(dev, tmp_path):
    """stable_jobs/wav2vec2_dist.py (T:1271-1339): banner, step lines, jct file, the model-saved line."""
    ws, res, out = _run_pair("stable_w2v", tmp_path)
    losses = {}
    for t in ("chief", "worker"):
        rc, txt = out[t]
        assert rc == 0
        lines = txt.split("\n")
        head = ["batch size per replica: 2, global batch size: 4", "num_batches: 3", "Wav2Vec2 분산 학습 시작..."] + BANNER
        tail, jct, losses[t] = _check_common(lines, head, 3)
        assert tail == [f"모델이 {os.path.join(ws, 'model_cache', 'wav2vec2_model')}에 저장되었습니다.", ""]
        assert open(os.path.join(res, "jobname", f"{t}_0_jct.txt")).read() == "%.2f" % jct
    assert losses["chief"] == losses["worker"]


def test_wav2vec2_single_main(dev, tmp_path, monkeypatch, capsys):
    """speech_jobs/wav2vec2_single.py (U:1279-1349): its own banner and closing lines, no result file."""
    sys.path.insert(0, os.path.join(ROOT, "speech_jobs"))
    ws = str(tmp_path / "w")
    os.makedirs(ws)
    monkeypatch.setenv("TETHYS_WORKSPACE", ws)
    monkeypatch.delenv("TF_CONFIG", raising=False)
    import wav2vec2_single
    rc = wav2vec2_single.main(["--batch_size", "2", "--num_batches", "3", "--model_size", "tiny"], model_overrides=V_OVER,
                              train_kw=dict(clip_samples=800))
    assert rc == 0
    lines = capsys.readouterr().out.split("\n")
    assert lines[:7] == ["Wav2Vec2 단일 GPU 학습 시작...", "선택된 모델 크기: tiny", "선택된 모델 타입: pretraining",
                         "Tiny 모델: 약 15-20M 파라미터", "모델 가중치 초기화 중...", "모델 가중치 초기화 완료", "에포크 1/1"]
    for i, l in enumerate(lines[7:10]):
        m = STEP_RE.match(l)
        assert m and int(m.group(1)) == i, l
    assert lines[10] == "학습 완료." and re.match(r"^JCT: \d+\.\d+$", lines[11])
    assert lines[12:] == [f"Tiny pretraining 모델이 {os.path.join(ws, 'model_cache', 'wav2vec2_tiny_pretraining_model')}에 저장되었습니다.", ""]
    with pytest.raises(SystemExit):
        wav2vec2_single.main(["--model_type", "asr"])

