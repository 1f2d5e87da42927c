"""Where a stage of the split-bf16 kernel-gradient product goes (csrc/wgrad_bf16.hip built with -DWB_STAMPS): shader-clock
stamps of every wave of workgroup (0, 0) at the top of each 32-row stage and at its arrival at the stage barrier.
  bash tools/build_variant.sh wbstamps "-DWB_STAMPS -fno-slp-vectorize" wgrad_bf16.hip
  CLV_LIB=$PWD/abtest/wbstamps/libclvae_hip.so python tools/wgrad_stamps.py [K] [nz]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd import _lib, ops  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T, N, nx, nh = 128, 352, 88, 88
dev = torch.device('cuda:0')
ldx = 88 if nz == 0 else 120
XZ = (torch.rand(K, ldx, device=dev) < 0.0443).float()
if nz:
    XZ[:, nx:nx + nz] = torch.randn(K, nz, device=dev)
hs, dz = torch.tanh(torch.randn(K, nh, device=dev)), torch.randn(K, N, device=dev)
gx, gu = torch.zeros(nx + nz, N, device=dev), torch.zeros(nh, N, device=dev)
ws, rq = ops.Workspace(dev), ops.ReduceQueue(dev)
fn = _lib.lib().clv_debug_wb_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
runs = []
for it in range(6):
    ops.lstm_wgrad(K, N, XZ, ldx, nx, True, hs, nh, nh, T, XZ[:, nx:] if nz else None, ldx, nz, dz, gx, gu, gx[nx:] if nz else None,
                   ws, defer=rq)
    rq.n = 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 1024)()
    assert fn(buf) == 0
    runs.append(np.array(buf[:], dtype=np.float64).reshape(64, 8, 2))
a = np.array(runs[2:])                                   # [launch, stage, wave, (top, at barrier)]
nst = int(min(64, (K // 128 + 31) // 32))
a = a[:, 1:nst - 1]                                      # interior stages
work = np.median(a[..., 1] - a[..., 0], axis=(0, 1))
last = a[..., 1].max(axis=2, keepdims=True)
wait = np.median(last[..., 0][..., None] - a[..., 1], axis=(0, 1))
stage = np.median(np.diff(a[:, :, :, 0], axis=1), axis=(0, 1))
print("K %d nz %d: %d stages per workgroup; cycles, median over interior stages (waves 0-3 issue MFMAs, 4-7 move data)" % (K, nz, nst))
print("%-34s" % "wave" + "".join("%8d" % w for w in range(8)))
print("%-34s" % "top of stage -> at the barrier" + "".join("%8.0f" % v for v in work))
print("%-34s" % "waits there for the last wave" + "".join("%8.0f" % v for v in wait))
print("%-34s" % "stage (top to top)" + "".join("%8.0f" % v for v in stage))
