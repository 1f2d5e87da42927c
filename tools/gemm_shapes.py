"""Time the GEMM shapes of a config-3 training step one by one (200 launches each, HIP events on the launch stream).
Usage (GPU box): python tools/gemm_shapes.py [path/to/libclvae_hip.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import _lib, ops

if len(sys.argv) > 1:           # another build of the library (A/B inside one run of the GPU box)
    _lib.LIB_PATH = sys.argv[1]

dev = torch.device('cuda:0')
f = lambda *s: torch.randn(*s, device=dev)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


BT, H = 32768, 88
ws = ops.Workspace(dev)
A, Wo, out = f(BT, H), f(H, H), f(BT, H)
print("NN  32768x88x88           %.1f us" % timeit(lambda: ops.gemm(A, Wo, out, BT, H, H, ws=ws)))
print("NT  32768x88x88           %.1f us" % timeit(lambda: ops.gemm(A, Wo, out, BT, H, H, tb=True, ws=ws)))
bias, Y, dl, rn = f(H), (torch.rand(BT, H, device=dev) < 0.05).float(), f(BT, H), f(BT)
print("NN+bce 32768x88x88        %.1f us" % timeit(lambda: ops.gemm_bce(A, Wo, bias, Y, 0.1, out, dl, rn, BT, H, H)))
dz, hs, X = f(BT, 352), f(BT, H), f(BT, H)
g1, g2 = f(H, 352), f(H, 352)
for rows in (88, 90):
    gx = f(rows, 352)
    Xr = f(BT, 92)
    print("TN grouped [%d + 88]x352x32768 %.1f us" % (rows, timeit(lambda: ops.gemm_grouped_tn(
        [dict(A=Xr, lda=92, M=rows, C=gx), dict(A=hs, lda=H, M=H, C=g2, shift=1, zero_period=128)], 352, BT, dz, ws))))
print("TN grouped [88]x352x32768      %.1f us" % timeit(lambda: ops.gemm_grouped_tn(
    [dict(A=hs, lda=H, M=H, C=g2, shift=1, zero_period=128)], 352, BT, dz, ws)))
gk = f(H + 1, H)
print("TN [88+ones]x88x32768          %.1f us" % timeit(lambda: ops.gemm_grouped_tn(
    [dict(A=hs, lda=H, M=H + 1, C=gk, ldc=H, ones=2)], H, BT, A, ws)))
