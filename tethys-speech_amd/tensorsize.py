"""Tensor-size / skewness report in the format of the reference's scheduler-research profiler
(speech_jobs/whisper_dist_tensorsize.py:20-458, driven at :1584-1586 and :1604-1613).

The reference sums ``tf.size * dtype.size`` over the tensors its layers log (TensorLoggingMixin,
:461-476; the names below are the ones logged at :595-777), the gradients (:100-105) and, once
before training, the parameters (:107-129), and derives the "Tiresias tensorsize" (mean step total
after a warm-up of min(3, n // 4) steps, :207-222) and the skewness of the size distribution
(scipy.stats.skew, :224-244).  Every one of those sizes is a function of the model configuration
and the batch shape alone, so here they are computed from the parameter arena and the shapes — no
instrumentation in the step, no runtime cost — and written to the same files:

  tensor_sizes.txt          step,operation,tensor_type,size_bytes,size_mb,shape
  summary.txt               step,total_tensor_size_mb,num_operations,avg_tensor_size_mb
  tiresias_tensorsize.txt   step,tensorsize_mb
  memory_usage.txt          step,gpu_memory_mb,cpu_memory_mb
  final_summary.json, tiresias_result.json, legacy_skewness_result.txt

``dtype_bytes`` is the element size tensors are counted at: 4 (the reference computes in fp32) by
default; pass 2 for what the bf16 path actually materialises.  Scores / probabilities are counted as
the reference materialises them ([B, H, Tq, Tk]) although the fused attention here never does.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Tuple

import numpy as np

MB = 1024 * 1024


def skew(values) -> float:
    """scipy.stats.skew with its defaults (biased Fisher-Pearson g1); 0.0 below 3 samples, as the
    reference's guard (:232)."""
    x = np.asarray(list(values), dtype=np.float64)
    if x.size < 3:
        return 0.0
    d = x - x.mean()
    m2 = np.mean(d * d)
    if m2 == 0.0:
        return 0.0
    return float(np.mean(d ** 3) / m2 ** 1.5)


def whisper_step_tensors(cfg, B: int, T: int, S: int) -> List[Tuple[str, str, Tuple[int, ...]]]:
    """(operation, tensor_type, shape) of every activation one training step logs, in call order:
    encoder positional encoding, then per encoder layer attention + feed-forward, decoder positional
    encoding, per decoder layer self-attention, cross-attention, feed-forward."""
    d, ff = cfg.d_model, cfg.d_ff
    He, Hd = cfg.encoder_attention_heads, cfg.decoder_attention_heads
    out: List[Tuple[str, str, Tuple[int, ...]]] = []

    def pe(rows):
        out.append(("positional_encoding_input", "activation", (B, rows, d)))
        out.append(("positional_encoding_output", "activation", (B, rows, d)))

    def attn(Tq, Tk, H, kind):
        hd = d // H
        out.append(("attention_hidden_states_input", "activation", (B, Tq, d)))
        for n in (f"{kind}_key_states", f"{kind}_value_states"):
            out.append((n, "activation", (B, H, Tk, hd)))
        out.append(("attention_query_states", "activation", (B, H, Tq, hd)))
        out.append(("attention_scores", "activation", (B, H, Tq, Tk)))
        if kind == "self_attention" and H == Hd and Tq == S and Tk == S:
            out.append(("attention_mask", "activation", (1, 1, Tq, Tk)))
        out.append(("attention_probs", "activation", (B, H, Tq, Tk)))
        out.append(("attention_output_raw", "activation", (B, H, Tq, hd)))
        out.append(("attention_output_final", "activation", (B, Tq, d)))

    def ffn(rows):
        out.append(("feedforward_input", "activation", (B, rows, d)))
        out.append(("feedforward_fc1_output", "activation", (B, rows, ff)))
        out.append(("feedforward_activation_output", "activation", (B, rows, ff)))
        out.append(("feedforward_fc2_output", "activation", (B, rows, d)))
        out.append(("feedforward_final_output", "activation", (B, rows, d)))

    pe(T)
    for _ in range(cfg.encoder_layers):
        attn(T, T, He, "self_attention")
        ffn(T)
    pe(S)
    for _ in range(cfg.decoder_layers):
        attn(S, S, Hd, "self_attention")
        attn(S, T, Hd, "cross_attention")
        ffn(S)
    return out


class TensorSizeReport:
    """Same files and summary keys as the reference's TensorProfiler, filled analytically."""

    def __init__(self, model, batch_size: int, seq_len: int = 3000, target_len: int = 100,
                 log_dir: str = "tensor_logs", dtype_bytes: int = 4, model_label: str = "whisper_small"):
        from .whisper import same_pad
        self.model, self.cfg = model, model.config
        self.B, self.S = int(batch_size), int(target_len)
        t1 = same_pad(seq_len, 3, 1)[0]
        self.T = same_pad(t1, 3, 2)[0]
        self.es = int(dtype_bytes)
        self.log_dir, self.label = log_dir, model_label
        a = model.arena
        self.variables: Dict[str, Tuple[int, ...]] = {k: tuple(v.shape) for k, v in a.ref_views(a.p).items()}
        self.step_tensor_sizes: List[float] = []
        self.tensor_details: List[dict] = []
        self.operation_tensor_sizes: Dict[str, List[int]] = {}
        os.makedirs(log_dir, exist_ok=True)
        self._f = {n: open(os.path.join(log_dir, n), "w") for n in
                   ("tensor_sizes.txt", "summary.txt", "tiresias_tensorsize.txt", "memory_usage.txt")}
        self._f["tensor_sizes.txt"].write("step,operation,tensor_type,size_bytes,size_mb,shape\n")
        self._f["summary.txt"].write("step,total_tensor_size_mb,num_operations,avg_tensor_size_mb\n")
        self._f["tiresias_tensorsize.txt"].write("step,tensorsize_mb\n")
        self._f["memory_usage.txt"].write("step,gpu_memory_mb,cpu_memory_mb\n")

    # -- one step --------------------------------------------------------------------------
    def _log(self, step, name, ttype, shape) -> int:
        size = int(np.prod(shape)) * self.es
        self.tensor_details.append({"step": step, "operation": name, "tensor_type": ttype, "size_bytes": size,
                                    "size_mb": size / MB, "shape": list(shape)})
        self.operation_tensor_sizes.setdefault(name, []).append(size)
        self._f["tensor_sizes.txt"].write(f"{step},{name},{ttype},{size},{size / MB:.4f},{list(shape)}\n")
        return size

    def _end(self, step, total, n_ops):
        mb = total / MB
        self.step_tensor_sizes.append(mb)
        self._f["summary.txt"].write(f"{step},{mb:.4f},{n_ops},{(mb / n_ops if n_ops else 0):.4f}\n")
        self._f["tiresias_tensorsize.txt"].write(f"{step},{mb:.4f}\n")
        return mb

    def log_parameters(self, step: int = 0) -> float:
        """The reference's pre-training pass (:1584-1586)."""
        total = sum(self._log(step, f"param_{k}", "parameter", s) for k, s in self.variables.items())
        return self._end(step, total, len(self.variables))

    def log_step(self, step: int) -> float:
        """One training step (:1604-1613): the layers' activations, then every gradient."""
        acts = whisper_step_tensors(self.cfg, self.B, self.T, self.S)
        total = sum(self._log(step, n, t, s) for n, t, s in acts)
        total += sum(self._log(step, f"gradient_{k}", "gradient", s) for k, s in self.variables.items())
        gpu = cpu = 0.0
        try:
            import torch
            if self.model.device.type == "cuda":
                gpu = torch.cuda.memory_allocated(self.model.device) / MB
            import psutil
            cpu = psutil.Process().memory_info().rss / MB
        except Exception:
            pass
        self._f["memory_usage.txt"].write(f"{step},{gpu:.2f},{cpu:.2f}\n")
        return self._end(step, total, len(acts) + len(self.variables))

    # -- summaries -------------------------------------------------------------------------
    def tiresias_tensorsize(self) -> float:
        s = self.step_tensor_sizes
        if not s:
            return 0.0
        warm = min(3, len(s) // 4)
        stable = s[warm:]
        return float(np.mean(stable if stable else s))

    def summary(self) -> dict:
        sizes = [t["size_mb"] for t in self.tensor_details if t["size_bytes"] > 0]
        by_type: Dict[str, List[float]] = {}
        for t in self.tensor_details:
            if t["size_mb"] > 0:
                by_type.setdefault(t["tensor_type"], []).append(t["size_mb"])
        return {
            "tiresias_tensorsize_mb": self.tiresias_tensorsize(),
            "model_skewness": skew(sizes),
            "layer_type_skewness": {k: skew(v) for k, v in by_type.items() if len(v) >= 3},
            "operation_skewness": {k: skew([x / MB for x in v]) for k, v in self.operation_tensor_sizes.items() if len(v) >= 3},
            "total_steps": len(self.step_tensor_sizes),
            "avg_step_size_mb": float(np.mean(self.step_tensor_sizes)) if self.step_tensor_sizes else 0.0,
            "max_step_size_mb": float(np.max(self.step_tensor_sizes)) if self.step_tensor_sizes else 0.0,
            "min_step_size_mb": float(np.min(self.step_tensor_sizes)) if self.step_tensor_sizes else 0.0,
            "dtype_bytes": self.es,
        }

    def save_final_results(self) -> dict:
        for f in self._f.values():
            f.flush()
        s = self.summary()
        with open(os.path.join(self.log_dir, "final_summary.json"), "w") as f:
            json.dump(s, f, indent=2, default=str)
        with open(os.path.join(self.log_dir, "tiresias_result.json"), "w") as f:
            json.dump({"model": self.label, "tensorsize_mb": s["tiresias_tensorsize_mb"], "skewness": s["model_skewness"],
                       "total_steps": s["total_steps"], "measurement_method": "Tiresias_style"}, f, indent=2)
        with open(os.path.join(self.log_dir, "legacy_skewness_result.txt"), "w") as f:
            f.write("model,skewness\n")
            f.write(f"{self.label},{s['model_skewness']:.1f}\n")
        return s

    def close(self):
        for f in self._f.values():
            f.close()
