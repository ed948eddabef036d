"""Is the forward bit-reproducible?  The same model and batch, forward_backward repeated: the loss (a function of the forward
only) printed as raw fp32 bits, for both precisions, with the second stream on and off."""
import json, os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper
from oracle import whisper_oracle as O
dev = "cuda:0"
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "whisper_small_ref_b8_10steps.json")))
params = O.init_params(O.make_config("small"), seed=gold["seed"], dtype=torch.float32)
feats, labels = O.create_dummy_pool(seed=gold["seed"])
f, l = next(O.batches(feats, labels, gold["batch_size"]))
f, l = torch.from_numpy(np.ascontiguousarray(f)).to(dev), torch.from_numpy(np.ascontiguousarray(l)).to(dev)
for prec in ("bf16", "fp32"):
    for rep in range(2):
        model = whisper.create_whisper_model("small", device=dev, precision=prec)
        model.arena.load_ref(params)
        model.refresh_shadows()
        out = []
        for i in range(5):
            v = float(model.forward_backward(f, l).item())
            out.append(struct.pack(">f", v).hex())
        print(prec, "model", rep, " ".join(out), flush=True)
        del model
        torch.cuda.empty_cache()
