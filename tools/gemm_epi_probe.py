"""The step's multi-round GEMMs (more tiles than CUs) WITH the epilogues they carry in the step: fc1 forward = bias + GELU +
saved pre-activation, fc2 dgrad = GELU' of the saved pre-activation, qkv forward = bias + q scale, LM head; plain forms beside
them.  Prints time per launch and the max error against torch (fp32 reference of the same bf16 operands).  Environment
switches of the library are read once per process: run once per setting (tools/gemm_epi_ab.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tethys_speech_amd  # noqa: F401
from tethys_speech_amd import ops
dev, bf = "cuda:0", torch.bfloat16
tag = os.environ.get("PROBE_TAG", "")
only = os.environ.get("PROBE_ONLY")


def timed(fn, iters=40):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def gelu(x):
    return torch.nn.functional.gelu(x, approximate="none")


def gelu_grad(u):
    return 0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5


torch.manual_seed(0)
cases = []
M = 12000
for name, N, K, layout, epi in [("fc1 fwd plain", 3072, 768, "nn", "plain"), ("fc1 fwd bias+gelu+aux", 3072, 768, "nn", "gelu"),
                                ("fc2 dgrad plain", 3072, 768, "nt", "plain"), ("fc2 dgrad gelu'", 3072, 768, "nt", "gelup"),
                                ("qkv fwd bias+scale", 2304, 768, "nn", "qkv"), ("fc2 fwd bias+resid", 768, 3072, "nn", "resid"),
                                ("fc1 dgrad plain", 768, 3072, "nt", "plain"), ("qkv dgrad plain", 768, 2304, "nt", "plain"),
                                ("out dgrad plain", 768, 768, "nt", "plain"), ("lm head", 51904, 768, "nn", "plain")]:
    if only and only not in name:
        continue
    Mm = 800 if name == "lm head" else M
    A = torch.randn(Mm, K, device=dev).to(bf)
    if layout == "nn":
        Bm = (torch.randn(K, N, device=dev) * 0.03).to(bf)
        b_sk, b_sn = N, 1
        Bf = Bm.float()
    else:
        Bm = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        b_sk, b_sn = 1, K
        Bf = Bm.float().t()
    Cm = torch.empty(Mm, N, device=dev, dtype=bf)
    bias = torch.randn(N, device=dev) * 0.1
    aux = torch.empty(Mm, N, device=dev, dtype=bf)
    u = torch.randn(Mm, N, device=dev).to(bf)
    res = torch.randn(Mm, N, device=dev).to(bf)
    kw = {}
    if epi == "gelu":
        kw = dict(bias=bias, act=1, aux_out=aux)
    elif epi == "gelup":
        kw = dict(aux_in=u)
    elif epi == "qkv":
        kw = dict(bias=bias, scale_cols=768, scale=0.125)
    elif epi == "resid":
        kw = dict(bias=bias, resid=res, r_ld=N)
    f = lambda: ops.gemm(A, Bm, Cm, Mm, N, K, K, 1, b_sk, b_sn, N, **kw)
    f()
    torch.cuda.synchronize()
    # reference on a row sample (every 37th row + the last 300: edge tiles included)
    rows = torch.cat([torch.arange(0, Mm, 37, device=dev), torch.arange(max(0, Mm - 300), Mm, device=dev)]).unique()
    ref = A[rows].float() @ Bf
    err_aux = 0.0
    if epi == "gelu":
        pre = ref + bias
        err_aux = float((aux[rows].float() - pre).abs().max())
        ref = gelu(pre.to(bf).float())  # the kernel applies GELU to the value it saved (rounded to bf16)
    elif epi == "gelup":
        ref = ref * gelu_grad(u[rows].float())
    elif epi == "qkv":
        ref = ref + bias
        ref[:, :768] *= 0.125
    elif epi == "resid":
        ref = ref + bias + res[rows].float()
    err = float((Cm[rows].float() - ref).abs().max())
    scale = float(ref.abs().max())
    t = timed(f)
    print(f"{tag:28s} {name:24s} {str((Mm, N, K)):>20s} {t:8.1f} us {2.0 * Mm * N * K / t / 1e6:7.1f} TF/s  max|err| {err:.3e} (|ref| {scale:.1f})"
          + (f" aux {err_aux:.2e}" if epi == "gelu" else ""), flush=True)
