"""Tensor-size / skewness report: analytic sizes, Tiresias mean, scipy-equal skew."""
import json
import os
import types

import numpy as np
import pytest
from scipy import stats

import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import tensorsize as TS
from tethys_speech_amd.whisper import make_config


def test_skew_matches_scipy():
    rng = np.random.default_rng(0)
    x = rng.gamma(2.0, 3.0, 500)
    assert abs(TS.skew(x) - stats.skew(x)) < 1e-12
    assert TS.skew([1.0, 2.0]) == 0.0 and TS.skew([5.0] * 10) == 0.0


def _fake_model(cfg):
    """arena stand-in: the report only needs names -> shapes and a device."""
    import torch
    from oracle import whisper_oracle as O
    shapes = O.param_shapes(O.make_config("small"))
    views = {k: types.SimpleNamespace(shape=tuple(v)) for k, v in shapes.items()}
    arena = types.SimpleNamespace(p=None, ref_views=lambda _p: views)
    return types.SimpleNamespace(config=cfg, arena=arena, device=torch.device("cpu"))


def test_report_files_and_totals(tmp_path):
    cfg = make_config("small")
    rep = TS.TensorSizeReport(_fake_model(cfg), batch_size=8, log_dir=str(tmp_path))
    assert rep.T == 1500
    p_mb = rep.log_parameters(0)
    assert abs(p_mb - 147_781_632 * 4 / TS.MB) < 1e-6
    sizes = [rep.log_step(s) for s in range(1, 9)]
    assert len(set(sizes)) == 1  # every step logs the same tensors
    acts = TS.whisper_step_tensors(cfg, 8, 1500, 100)
    act_bytes = sum(int(np.prod(s)) for _, _, s in acts) * 4
    assert abs(sizes[0] * TS.MB - (act_bytes + 147_781_632 * 4)) < 1.0
    # scores are the big ones: [8, 12, 1500, 1500] fp32 = 864 MB each
    assert max(int(np.prod(s)) for _, _, s in acts) == 8 * 12 * 1500 * 1500
    s = rep.save_final_results()
    rep.close()
    # Tiresias mean: 9 logged steps (params + 8), warm-up min(3, 9 // 4) = 2 dropped
    assert abs(s["tiresias_tensorsize_mb"] - sizes[0]) < 1e-9 and s["total_steps"] == 9
    for f in ("tensor_sizes.txt", "summary.txt", "tiresias_tensorsize.txt", "memory_usage.txt", "final_summary.json",
              "tiresias_result.json", "legacy_skewness_result.txt"):
        assert os.path.exists(tmp_path / f), f
    assert open(tmp_path / "tensor_sizes.txt").readline().strip() == "step,operation,tensor_type,size_bytes,size_mb,shape"
    res = json.load(open(tmp_path / "tiresias_result.json"))
    assert res["measurement_method"] == "Tiresias_style" and res["model"] == "whisper_small"
    rows = [l.split(",")[3] for l in open(tmp_path / "tensor_sizes.txt").read().splitlines()[1:]]
    mb = [int(r) / TS.MB for r in rows if int(r) > 0]
    assert abs(s["model_skewness"] - stats.skew(mb)) < 1e-9
