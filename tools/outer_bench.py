"""The hW layer's kernel gradient alone: the note-walking kernel (sparse_outer) against the dense bf16 product
(dense_outer_bf16) at the configuration-3 and configuration-5 shapes.  Usage (GPU box): python tools/outer_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import ops

dev = torch.device('cuda:0')


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


for B, T in ((256, 128), (1024, 256), (512, 128), (128, 128)):
    nx, N = T * 88, 88
    X = (torch.rand(B, nx, device=dev) < 0.0443).float()
    G, H, hb = torch.randn(B, N, device=dev), torch.relu(torch.randn(B, N, device=dev)), torch.randn(N, device=dev)
    out, cs, gd = torch.empty(nx, N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
    ts = timeit(lambda: ops.sparse_outer(B, nx, N, X, nx, G, N, out, colsum=cs, gdot=(H, N, hb, gd)))
    td = timeit(lambda: ops.dense_outer_bf16(B, nx, N, X, nx, G, N, out, colsum=cs, gdot=(H, N, hb, gd)))
    print("B=%4d T=%3d: sparse_outer %6.1f us   dense_outer_bf16 %6.1f us   (X %.0f MB)" % (B, T, ts, td, B * nx * 4 / 1e6))
