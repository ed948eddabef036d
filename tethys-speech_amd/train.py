"""Training step and loop with the reference's surface (W:819-848, W:894-958)."""
from __future__ import annotations

import os
import time

import torch

from .data import create_dummy_dataset
from .dist import DataParallelStrategy
from .optim import Adam
from .whisper import create_whisper_model


def distributed_train_step(strategy, model, dist_inputs, optimizer):
    """W:819-848.  Per replica: forward, backward, apply_gradients (all-reduce SUM, then
    Adam); returns ``strategy.reduce(SUM, per_replica_loss)`` as a 1-element device tensor.
    A replica whose slice of a short final batch is empty contributes zero gradients."""
    features, labels = dist_inputs
    strategy.begin_gradients(model.arena.g)
    if features.shape[0] > 0:
        loss = model.forward_backward(features, labels, grad_ready=strategy.gradients_ready)
    else:
        model.arena.g.zero_()
        loss = torch.zeros(1, dtype=torch.float32, device=model.device)
    optimizer.apply_gradients(model, strategy)
    return strategy.reduce_sum(loss.clone())


def train_whisper(strategy, model_type="small", num_epochs=1, learning_rate=1e-4, *, batch_size=1,
                  num_batches=40, precision="bf16", device="cuda:0", checkpoint_dir=None, log=print, seed=1234,
                  model_overrides=None, seq_len=3000, max_target_length=100):
    """W:894-958: model + Adam(1e-4), dummy dataset, per-step log line, checkpoint at epoch end."""
    model = create_whisper_model(model_type, device=device, precision=precision, seed=seed,
                                 **(model_overrides or {}))
    strategy.broadcast_parameters(model.arena.p)
    model.refresh_shadows()
    optimizer = Adam(learning_rate=learning_rate)
    ds = create_dummy_dataset(batch_size, n_mels=model.config.n_mels, seq_len=seq_len,
                              max_target_length=max_target_length, device=device, rank=strategy.rank,
                              world=strategy.world, seed=seed, drop_remainder=strategy.world > 1)
    it = iter(ds)
    step = 0
    losses = []
    start_time = time.time()
    for epoch in range(num_epochs):
        log(f"Epoch {epoch + 1}/{num_epochs}")
        for _ in range(num_batches):
            inputs = next(it)
            step_start = time.time()
            loss = distributed_train_step(strategy, model, inputs, optimizer)
            lv = float(loss.item())  # the reference's loss.numpy() host sync (W:951)
            step_end = time.time()
            losses.append(lv)
            log(f"Step {step}, Loss: {lv:.4f}, Time: {time.strftime('%H:%M:%S')} "
                f"(경과: {step_end - start_time:.2f}초, 스텝 시간: {step_end - step_start:.2f}초)")
            step += 1
        if checkpoint_dir and strategy.rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            save_checkpoint(model, optimizer, os.path.join(checkpoint_dir, f"whisper_{model_type}_epoch_{epoch + 1}.pt"))
    model.losses = losses
    return model


def save_checkpoint(model, optimizer, path):
    """tf.train.Checkpoint(model, optimizer).save (W:919,956): flat-arena dump."""
    a = model.arena
    torch.save({"p": a.p.cpu(), "m": a.m.cpu(), "v": a.v.cpu(), "iterations": optimizer.iterations,
                "names": a.names, "offsets": a.offsets, "shapes": a.shapes}, path)


def load_checkpoint(model, optimizer, path):
    """The restore path the reference lacks."""
    ck = torch.load(path, map_location="cpu")
    a = model.arena
    if ck["names"] != a.names:
        raise ValueError("checkpoint layout does not match the model")
    a.p.copy_(ck["p"]); a.m.copy_(ck["m"]); a.v.copy_(ck["v"])
    optimizer.iterations = int(ck["iterations"])
    model.refresh_shadows()
