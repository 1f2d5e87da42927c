"""Debug aid for csrc/out_head_bf16.hip: per-output errors of clv_out_head_train against fp64 numpy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import clvae_amd  # noqa
from clvae_amd import ops
dev = torch.device('cuda:0')
T = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
for R in (1, 16, 37, 128, 1000):
    rng = np.random.default_rng(R)
    H = D = 88
    hs = np.tanh(rng.standard_normal((R, H))).astype(np.float32)
    Wo = (rng.standard_normal((H, D)) * 0.4).astype(np.float32)
    bo = rng.standard_normal(D).astype(np.float32)
    Y = (rng.random((R, D)) < 0.1).astype(np.float32)
    scale = 1.0 / R
    z = lambda *sh: torch.full(sh, -7.0, dtype=torch.float32, device=dev)
    logits, dl, rownll, dhs, dWo, dbo = z(R, D), z(R, D), z(R), z(R, H), z(H, D), z(D)
    ws = ops.Workspace(dev)
    ops.out_head_train(R, H, D, T(hs), T(Wo), T(bo), T(Y), scale, rownll, dhs, dWo, dbo, ws, logits=logits, dlogits=dl)
    torch.cuda.synchronize()
    a = hs.astype(np.float64) @ Wo.astype(np.float64) + bo
    nll = (np.maximum(a, 0) + np.log1p(np.exp(-np.abs(a))) - a * Y).sum(1)
    dlr = scale * (1 / (1 + np.exp(-a)) - Y)
    N = lambda t: t.cpu().numpy().astype(np.float64)
    e = lambda g, w: float(np.abs(g - w).max())
    print("R=%d logits %.2e dl %.2e nll %.2e dhs %.2e dWo %.2e dbo %.2e" % (
        R, e(N(logits), a), e(N(dl), dlr), e(N(rownll), nll), e(N(dhs), dlr @ Wo.astype(np.float64).T),
        e(N(dWo), hs.astype(np.float64).T @ dlr), e(N(dbo), dlr.sum(0))))
    if R == 1:
        d = np.abs(N(logits) - a)[0]
        print(" logits err by note:", np.round(d, 3))
