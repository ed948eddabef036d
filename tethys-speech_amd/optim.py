"""tf.keras.optimizers.Adam over the flat parameter arena (W:901; V:1271-1275).

One kernel launch for the whole model (``tmi_adam_step``): Keras-V2 epsilon placement by
default (``eps_mode="tf"``), AdamW-style decoupled weight decay available but 0 as in the
reference.  ``apply_gradients`` is the analogue of ``optimizer.apply_gradients`` (W:834):
it first lets the strategy all-reduce the gradient arena (the implicit cross-replica SUM
of Keras OptimizerV2), then updates; the same kernel writes the model's bf16 weight mirror.
"""
from __future__ import annotations

from . import ops


class Adam:
    def __init__(self, learning_rate=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7, eps_mode="tf",
                 weight_decay=0.0):
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        if eps_mode not in ("tf", "torch"):
            raise ValueError("eps_mode must be 'tf' or 'torch'")
        self.eps_mode = 0 if eps_mode == "tf" else 1
        self.weight_decay = float(weight_decay)
        self.iterations = 0

    def apply_gradients(self, model, strategy=None, grad_scale=1.0):
        a = model.arena
        if strategy is not None:
            strategy.all_reduce_gradients(a.g)
        self.iterations += 1
        ops.adam_step(a.p, a.g, a.m, a.v, a.numel, self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
                      self.iterations, self.eps_mode, self.weight_decay, grad_scale, mirror=model.mirror)

    # -- captured-graph form: the launch reads its step-dependent scalars from device memory
    def scalars(self, step=None):
        return ops.adam_scalars(self.learning_rate, self.beta_1, self.beta_2, self.iterations if step is None else step,
                                self.eps_mode, self.weight_decay)

    def apply_gradients_dev(self, model, dev_scalars, grad_scale=1.0):
        """The update of ``apply_gradients`` with [step_size, vcorr, decay] taken from ``dev_scalars``
        (a 3-float device tensor the caller refreshes from ``scalars()`` before every replay)."""
        a = model.arena
        ops.adam_step_dev(a.p, a.g, a.m, a.v, a.numel, self.beta_1, self.beta_2, self.epsilon, dev_scalars,
                          self.eps_mode, grad_scale, mirror=model.mirror)
