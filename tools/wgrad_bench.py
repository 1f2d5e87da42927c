"""The LSTM weight-gradient product alone: split-bf16 kernel (csrc/wgrad_bf16.hip) vs the grouped f32-MFMA GEMM, with
and without their split-K reductions.  Usage (GPU box): python tools/wgrad_bench.py [K] [T]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import ops

dev = torch.device('cuda:0')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N, nx, nh = 352, 88, 88


def timeit(fn, n=100):
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


for nz in (0, 2, 32):
    ldx = 88 if nz == 0 else (92 if nz <= 4 else 120)
    XZ = (torch.rand(K, ldx, device=dev) < 0.0443).float()
    if nz:
        XZ[:, nx:nx + nz] = torch.randn(K, nz, device=dev)
    hs, dz = torch.tanh(torch.randn(K, nh, device=dev)), torch.randn(K, N, device=dev)
    gx, gu, gz = torch.zeros(nx + nz, N, device=dev), torch.zeros(nh, N, device=dev), None
    ws, rq = ops.Workspace(dev), ops.ReduceQueue(dev)
    Z = XZ[:, nx:] if nz else None
    gzv = gx[nx:] if nz else None

    def bf16(defer, exact=True):
        ops.lstm_wgrad(K, N, XZ, ldx, nx, exact, hs, nh, nh, T, Z, ldx, nz, dz, gx, gu, gzv, ws, defer=rq if defer else None)
        if defer:
            rq.n = 0           # drop the pending reduction: kernel time only

    def f32(defer):
        ops.gemm_grouped_tn([dict(A=XZ, lda=ldx, M=nx + nz, C=gx), dict(A=hs, lda=nh, M=nh, C=gu, shift=1, zero_period=T)],
                            N, K, dz, ws, defer=rq if defer else None)
        if defer:
            rq.n = 0

    print("K=%d nz=%2d  bf16 kernel %6.1f us  +reduce %6.1f us | bf16 (3-piece frames) %6.1f us | f32 GEMM %6.1f us  +reduce %6.1f us"
          % (K, nz, timeit(lambda: bf16(True)), timeit(lambda: bf16(False)),
             timeit(lambda: bf16(True, False)) if ops.lstm_wgrad_supported(N, nx, nh, nz, False) else float('nan'),
             timeit(lambda: f32(True)), timeit(lambda: f32(False))))
