"""The training loops behind the CLI shims (W:894-958, V:1263-1376) run end to end on small models: per-step
log lines, checkpoints, dropout on by default on the bf16 path and off on request."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dropout", [None, False])
def test_whisper_training_loop(dev, tmp_path, dropout):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, train
    over = dict(d_model=128, encoder_attention_heads=2, decoder_attention_heads=2, d_ff=256, vocab_size=160,
                encoder_layers=2, decoder_layers=2, n_mels=16, n_ctx=64, decoder_start_token_id=150, max_target_positions=32)
    lines = []
    model = train.train_whisper(dist.DataParallelStrategy(0, 1), batch_size=4, num_batches=5, precision="bf16", device=dev,
                                checkpoint_dir=str(tmp_path), log=lines.append, model_overrides=over, seq_len=96,
                                max_target_length=12, dropout=dropout)
    steps = [l for l in lines if l.startswith("Step ")]
    assert len(steps) == 5 and all("Loss:" in l for l in steps)
    assert (model._drop_p > 0) == (dropout is None)  # default: the reference's rates on the bf16 path
    train.wait_for_checkpoints()  # (epoch-end checkpoints are written by a background thread)
    assert any(f.endswith(".pt") for f in os.listdir(tmp_path))
    assert all(torch.isfinite(torch.tensor(model.losses)))


@pytest.mark.parametrize("dropout", [None, False])
def test_wav2vec2_training_loop(dev, tmp_path, dropout):
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import dist, train
    over = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                conv_dim=(64, 64, 64), conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2), num_conv_pos_embeddings=8,
                num_conv_pos_embedding_groups=4, num_codevectors_per_group=16, codevector_dim=32,
                proj_codevector_dim=64, num_negatives=10)
    lines = []
    model = train.train_wav2vec2(dist.DataParallelStrategy(0, 1), model_size="base", batch_size=4, num_batches=4,
                                 precision="bf16", device=dev, checkpoint_dir=str(tmp_path), log=lines.append,
                                 clip_samples=800, model_overrides=over, dropout=dropout)
    steps = [l for l in lines if l.startswith("Step ")]
    assert len(steps) == 4
    assert (model._drop_p > 0) == (dropout is None)
    train.wait_for_checkpoints()  # (epoch-end checkpoints are written by a background thread)
    assert any(f.endswith(".pt") for f in os.listdir(tmp_path))
    assert all(torch.isfinite(torch.tensor(model.losses)))
