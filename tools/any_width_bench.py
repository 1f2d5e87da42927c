"""cl_vrnn training step at other --intermediate_dim values (csrc/lstm_any.hip + the generic chain) next to the default 88:
ms per step at the reference CLI's default shape (batch 200 x seq_length 16, latent 2, 10 classes).  Usage (GPU box):
python tools/any_width_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clvae_amd  # noqa: F401
from clvae_amd.engine import VrnnEngine
from clvae_amd.initializers import init_weights
from clvae_amd.trainer import TrainStep

dev = torch.device('cuda:0')
B, T, L, C = 200, 16, 2, 10
rng = np.random.default_rng(0)
win = torch.as_tensor((rng.random((4 * B, T + 1, 88)) < 0.0443).astype(np.uint8), device=dev)
keys = torch.as_tensor(np.eye(C, dtype=np.float32)[rng.integers(0, C, 4 * B)], device=dev)
for H, extra in ((88, {}), (88, dict(fuse_pair=False)), (32, {}), (64, {}), (128, {}), (256, {})):
    cfg = dict(D=88, H=H, L=L, T=T, C=C, use_x_prev=True, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0, w_log_var_prior=0.0,
               gate_act='hard_sigmoid', **extra)
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(init_weights(eng.P.logical, cfg, seed=0))
    ts = TrainStep(eng, seed=1)
    ts.bind_batches(win[:, 1:].reshape(4 * B, -1).contiguous(), win[:, :-1].reshape(4 * B, -1).contiguous(), keys, idx=None, period=4, stride=B)
    for _ in range(10):
        ts.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        ts.step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 200
    print("H %3d %-18s %.3f ms per step  %.1f M timesteps/s  (pair %s)" % (H, extra or '', ms, B * T / ms / 1e3, eng.fuse_pair))
