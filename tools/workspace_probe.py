"""Diagnostics: training steps with a guard zone behind every workspace buffer and NaN in every torch.empty one
(TMI_WS_GUARD / TMI_WS_POISON, see blocks.KernelBlocks._buf) at the BENCH sizes: Whisper (any --model_type) at B = 8 with
30 s clips, Wav2Vec2-base at B = 8 with 2 s and 5 s clips.  Prints the buffers a kernel ran past and whether the loss /
gradients stayed finite.  (tests/test_workspace_guards_gpu.py is the same check at test sizes.)
usage: workspace_probe.py [whisper|wav2vec2] [model_type/size] [batch]"""
import os, sys
os.environ.setdefault("TMI_WS_GUARD", "4096")
os.environ.setdefault("TMI_WS_POISON", "1")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import dist as D, optim, train, whisper, wav2vec2
from tethys_speech_amd.data import create_dummy_dataset, W2VDummyDataset

what = sys.argv[1] if len(sys.argv) > 1 else "whisper"
size = sys.argv[2] if len(sys.argv) > 2 else ("small" if what == "whisper" else "base")
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = "cuda:0"
strat = D.DataParallelStrategy(0, 1)


def report(model, loss, tag):
    lv = float(loss.item())
    over = model.check_workspace_guards()
    print(f"{tag}: loss {lv:.4f} finite={np.isfinite(lv)} overrun buffers: {over}", flush=True)


if what == "whisper":
    for drop in (False, True):
        model = whisper.create_whisper_model(size, device=dev, precision="bf16", seed=1)
        model.refresh_shadows()
        if drop:
            model.enable_dropout(0.1, 0.1, seed=3)
        opt = optim.Adam(1e-4)
        it = iter(create_dummy_dataset(B, device=dev, seed=1, drop_remainder=False))  # 50 clips: the 7th batch is short
        for s in range(8):
            loss = train.distributed_train_step(strat, model, next(it), opt)
            if s in (0, 6, 7):
                report(model, loss, f"whisper-{size} B={B} dropout={drop} step {s}")
        g = model.arena.g  # (zeroed by Adam) - parameters must be finite
        print("  params finite:", bool(torch.isfinite(model.arena.p).all()))
        del model, opt
        torch.cuda.empty_cache()
else:
    for clip in (32000, 80000):
        for drop in (False, True):
            model = wav2vec2.create_full_model("pretraining", size, device=dev, precision="bf16", seed=1)
            model.refresh_shadows()
            if drop:
                c = model.config
                model.enable_dropout(c.hidden_dropout, c.attention_dropout, seed=3, act_p=c.activation_dropout)
            opt = optim.Adam(3e-5, epsilon=1e-8)
            ds = W2VDummyDataset(B, length=clip, device=dev, seed=1, drop_remainder=False)
            it = iter(ds)
            rng = np.random.default_rng(0)
            for s in range(8):
                a = next(it)
                model._prepare(a.shape[0], clip)
                if clip == 32000:
                    model.neg_per_time = False
                    neg = torch.from_numpy(wav2vec2.sample_negative_indices(rng, a.shape[0], model.T, model.config.num_negatives)).to(dev)
                    loss = train.wav2vec2_train_step(strat, model, a, neg, opt)
                else:
                    neg = torch.from_numpy(wav2vec2.sample_negative_indices_roll(rng, model.T, model.config.num_negatives)).to(dev)
                    loss = train.single_train_step(model, a, neg, opt)
                if s in (0, 6, 7):
                    report(model, loss, f"wav2vec2-{size} B={a.shape[0]} clip={clip} dropout={drop} step {s}")
            print("  params finite:", bool(torch.isfinite(model.arena.p).all()))
            del model, opt
            torch.cuda.empty_cache()
