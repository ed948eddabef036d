"""Where the bf16 loss-curve error of the headline golden comes from (north star: 1e-3): the small-ref B = 8 10-step run of
tests/test_whisper_step_gpu.py::test_whisper_small_ref_b8_loss_curve_golden with SIGNED per-step errors, twice (run-to-run
noise of the fp32 atomics), under whatever library switches the environment sets; MARGIN_VARIANT picks a host-side variant:
  fp32_logits   the LM head writes fp32 logits (the loss is evaluated on unrounded logits)
  fp32_master   forward / backward GEMMs read weights re-rounded from the fp32 master every step (the default anyway)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper, optim, dist, train
from oracle import whisper_oracle as O
dev = "cuda:0"
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "whisper_small_ref_b8_10steps.json")))
tag = os.environ.get("PROBE_TAG", "")
for rep in range(int(os.environ.get("MARGIN_REPS", "2"))):
    ocfg = O.make_config("small")
    params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
    model = whisper.create_whisper_model("small", device=dev, precision="bf16")
    model.arena.load_ref(params)
    model.refresh_shadows()
    feats, labels = O.create_dummy_pool(seed=gold["seed"])
    opt = optim.Adam(learning_rate=gold["lr"])
    strat = dist.DataParallelStrategy(0, 1)
    it = O.batches(feats, labels, gold["batch_size"])
    got = []
    for _ in range(len(gold["losses"])):
        f, l = next(it)
        loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev),
                                                           torch.from_numpy(np.ascontiguousarray(l)).to(dev)), opt)
        got.append(float(loss.item()))
    err = [a - b for a, b in zip(got, gold["losses"])]
    print(f"{tag:36s} rep {rep}: max |dloss| {max(abs(e) for e in err):.2e}  signed x1e4: {' '.join('%+5.1f' % (e * 1e4) for e in err)}", flush=True)
    del model
    torch.cuda.empty_cache()
