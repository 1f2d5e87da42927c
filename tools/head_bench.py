"""Time clv_out_head_train at the config-3 size (R = 32768) against the three launches it replaces.
Usage (GPU box): python tools/head_bench.py [path/to/other/libclvae_hip.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import _lib, ops

if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
dev = torch.device('cuda:0')
f = lambda *s: torch.randn(*s, device=dev)


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


R, H = int(os.environ.get('R', 32768)), 88
ws = ops.Workspace(dev)
hs, Wo, bo = torch.tanh(f(R, H)), f(H, H) * 0.3, f(H)
Y = (torch.rand(R, H, device=dev) < 0.05).float()
logits, rn, dhs, dWo, dbo = f(R, H), f(R), f(R, H), f(H, H), f(H)
print("%s: out_head_train R=%d  %.1f us" % (os.path.basename(os.path.dirname(_lib.LIB_PATH)) or '.', R, timeit(
    lambda: ops.out_head_train(R, H, H, hs, Wo, bo, Y, 1.0 / R, rn, dhs, dWo, dbo, ws, logits=logits))))
