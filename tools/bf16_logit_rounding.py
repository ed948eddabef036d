"""How much of the bf16 loss-curve error is the bf16 ROUNDING OF THE LOGITS (the loss is evaluated on logits stored in bf16)?
After every step of the headline golden's run: the decoder output and the LM-head weights the step used (both bf16) give fp32
logits (exact products, fp32 accumulation); the loss on them against the loss on the same logits rounded to bf16."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import whisper, optim, dist, train
from oracle import whisper_oracle as O
dev = "cuda:0"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gold = json.load(open(os.path.join(root, "tests", "golden", "whisper_small_ref_b8_10steps.json")))
ocfg = O.make_config("small")
params = O.init_params(ocfg, seed=gold["seed"], dtype=torch.float32)
model = whisper.create_whisper_model("small", device=dev, precision="bf16")
model.arena.load_ref(params)
model.refresh_shadows()
feats, labels = O.create_dummy_pool(seed=gold["seed"])
opt = optim.Adam(learning_rate=gold["lr"])
strat = dist.DataParallelStrategy(0, 1)
it = O.batches(feats, labels, gold["batch_size"])
V = model.config.vocab_size
for step in range(len(gold["losses"])):
    f, l = next(it)
    lab = torch.from_numpy(np.ascontiguousarray(l)).to(dev)
    wl, ldw = model.W("lm_head.kernel")
    w_before = wl.clone()                      # the weights this step's forward reads
    loss = train.distributed_train_step(strat, model, (torch.from_numpy(np.ascontiguousarray(f)).to(dev), lab), opt)
    x = model.ws["dec_out"]                    # [B*S, d] bf16, still this step's
    B, S = lab.shape
    logits32 = (x.float() @ w_before.float().view(-1, ldw))[:, :V].view(B, S, V)
    tgt = lab[:, 1:].long().reshape(-1)
    ce = lambda lg: float(torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, V).double(), tgt))
    l32, l16 = ce(logits32), ce(logits32.bfloat16().float())
    print(f"step {step}: step loss {float(loss):.6f} (golden {gold['losses'][step]:.6f}, err {(float(loss) - gold['losses'][step]) * 1e4:+5.1f}e-4)  "
          f"fp32-logit loss {l32:.6f}  bf16-logit loss {l16:.6f}  rounding of the logits {(l16 - l32) * 1e4:+5.2f}e-4  "
          f"fp32-logit loss - golden {(l32 - gold['losses'][step]) * 1e4:+5.1f}e-4", flush=True)
