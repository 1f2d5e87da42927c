"""Fixed (per-launch) vs per-step cost of the pair kernels: event times at several sequence lengths, batch 256.
Usage (GPU box): python tools/pair_fixed_cost.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clvae_amd  # noqa: F401
from clvae_amd import _lib, ops

if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from clvae_amd.engine import VrnnEngine
from oracle import clvae_oracle as O

dev = torch.device('cuda:0')
B = 256
for T in (1, 2, 8, 32, 128):
    cfg = O.vrnn_config(latent_dim=2, seq_length=T, n_classes=10, use_x_prev=True)
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights({k: np.asarray(v, dtype=np.float32) for k, v in O.vrnn_init_params(cfg, seed=1).items()})
    rng = np.random.default_rng(0)
    win = (rng.random((B, T + 1, 88)) < 0.0443).astype(np.float32)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
    args = (t(win[:, 1:]), t(win[:, :-1]), t(np.eye(10)[rng.integers(0, 10, B)]), t(rng.standard_normal((B, 9))),
            t(rng.standard_normal((B, T, 2))))
    for _ in range(5):
        eng.loss_and_grads(*args)
    torch.cuda.synchronize()
    ops.prof_enable(True)
    for _ in range(50):
        eng.loss_and_grads(*args)
    recs = {r[0]: r[2] / r[1] * 1e3 for r in ops.prof_collect()}
    ops.prof_enable(False)
    print("T=%3d  pair_fwd %7.1f us  pair_bwd %7.1f us" % (T, recs.get('lstm_pair_fwd', 0), recs.get('lstm_pair_bwd', 0)))
