"""Decoder-sized GEMMs (M = B*100 = 800 rows) under the tile configurations tmi_gemm can be forced to
(TMI_GEMM_CFG is read once per process: one child process per configuration)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    import tethys_speech_amd  # noqa: F401
    from tethys_speech_amd import ops
    dev = "cuda:0"; bf = torch.bfloat16

    def bench(fn, iters=100):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters
    out = []
    M = 800
    COLD = os.environ.get("GEMM_COLD", "0") != "0"   # weights rotate through > 256 MB of copies: every launch reads them from HBM, as in the step
    for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
        ncopy = max(1, (300 << 20) // (N * K * 2)) if COLD else 1
        X = torch.randn(M, K, device=dev).to(bf)
        Wts = [torch.randn(N, K, device=dev).to(bf) for _ in range(ncopy)]; Ws = [torch.randn(K, N, device=dev).to(bf) for _ in range(ncopy)]
        Y = torch.empty(M, N, device=dev, dtype=bf)
        ctr = [0]
        def fwd():
            ctr[0] += 1
            ops.gemm(X, Ws[ctr[0] % ncopy], Y, M, N, K, K, 1, N, 1, N)
        def dgrd():
            ctr[0] += 1
            ops.gemm(X, Wts[ctr[0] % ncopy], Y, M, N, K, K, 1, 1, K, N)
        W, Wt = Ws[0], Wts[0]
        out.append((f"fwd  KC,KS M{M} N{N} K{K}", bench(fwd)))
        out.append((f"dgrd KC,KC M{M} N{N} K{K}", bench(dgrd)))
        del Wts, Ws
    for Mo, No in ((768, 768), (768, 3072), (3072, 768), (768, 2304)):
        X = torch.randn(M, Mo, device=dev).to(bf); DY = torch.randn(M, No, device=dev).to(bf)
        G = torch.zeros(Mo, No, device=dev)
        out.append((f"wgrd KS,KS out {Mo}x{No} K{M}", bench(lambda: ops.gemm(X, DY, G, Mo, No, M, 1, Mo, No, 1, No, splitk=0))))
    for k, v in out:
        print(f"{k:36s} {v:7.1f}")
    # correctness of whatever configuration is forced, on ragged sizes too (stderr: the parent only parses timings)
    for (M2, N2, K2) in ((800, 768, 768), (800, 768, 3072), (333, 200, 448), (64, 64, 64), (130, 72, 1024)):
        X = torch.randn(M2, K2, device=dev).to(bf); Wt = torch.randn(N2, K2, device=dev).to(bf); W = torch.randn(K2, N2, device=dev).to(bf)
        Y = torch.empty(M2, N2, device=dev, dtype=bf)
        ops.gemm(X, W, Y, M2, N2, K2, K2, 1, N2, 1, N2)
        e1 = (Y.float() - X.float() @ W.float()).abs().max().item() / (X.float() @ W.float()).abs().max().item()
        ops.gemm(X, Wt, Y, M2, N2, K2, K2, 1, 1, K2, N2)
        e2 = (Y.float() - X.float() @ Wt.float().t()).abs().max().item() / (X.float() @ Wt.float().t()).abs().max().item()
        print(f"check cfg={os.environ.get('TMI_GEMM_CFG', 'auto')} M{M2} N{N2} K{K2}: rel err fwd {e1:.2e} dgrad {e2:.2e}", file=sys.stderr)
        assert e1 < 1e-2 and e2 < 1e-2
    sys.exit(0)
res = {}
cfgs = ["", "12", "13", "15", "11", "6", "4"]
for c in cfgs:
    env = dict(os.environ)
    if c:
        env["TMI_GEMM_CFG"] = c
    else:
        env.pop("TMI_GEMM_CFG", None)
    p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    for line in p.stdout.splitlines():
        if len(line) > 36 and line[36:].strip().replace(".", "").isdigit():
            res.setdefault(line[:36], {})[c or "auto"] = float(line[36:])
    if p.returncode:
        print("cfg", c, "failed:", p.stderr[-300:])
    elif c in ("15",):
        print(p.stderr.strip())
print(f"{'shape':36s} " + " ".join(f"{(c or 'auto'):>7s}" for c in cfgs) + "   (us; cfg 12 = 64x64 4 waves, 13 = + 4-stage ring, 15 = K-groups 2x4 waves, 4 = 128x128, 11 = 2 waves ring, 6 = 2 waves)")
for k, v in res.items():
    print(f"{k:36s} " + " ".join(f"{v.get(c or 'auto', float('nan')):7.1f}" for c in cfgs))
