"""Timeline of the workgroups of one lstm_wgrad_bf16 launch (csrc/wgrad_bf16.hip built with -DWB_STAMPS): the 100 MHz clock
of wave 0 at entry, LDS cleared, first stage ready, stages done, slab stored.
  bash tools/build_variant.sh wbstamps "-DWB_STAMPS -fno-slp-vectorize" wgrad_bf16.hip
  CLV_LIB=$PWD/abtest/wbstamps/libclvae_hip.so python tools/wgrad_wg_timeline.py [pair|single] [K]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
mode = sys.argv[1] if len(sys.argv) > 1 else 'pair'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
T, N, nx, nh = 128, 352, 88, 88
f = lambda *s: torch.randn(*s, device=dev)
X = (torch.rand(K, 88, device=dev) < 0.0443).float()
XZ = torch.zeros(K, 92, device=dev); XZ[:, :88] = X; XZ[:, 88:90] = f(K, 2)
hs, dz = torch.tanh(f(K, nh)), f(K, N)
ge, gue, gd, gud = f(nx, N), f(nh, N), f(nx + 2, N), f(nh, N)
pe = (K, N, X, 88, nx, True, hs, nh, nh, T, None, 88, 0, dz, ge, gue, None)
pd = (K, N, XZ, 92, nx, True, hs, nh, nh, T, XZ[:, 88:], 92, 2, dz, gd[:88], gud, gd[88:])
ws = ops.Workspace(dev)
fn = _lib.lib().clv_debug_wb_wg
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
for it in range(4):
    rq = ops.ReduceQueue(dev)
    if mode == 'pair':
        ops.lstm_wgrad_pair(pd, pe, None, defer=rq)
        nwg = 2 * 2 * 64
    else:
        ops.lstm_wgrad(*pd, ws, defer=rq)
        nwg = 2 * 128
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 5120)()
assert fn(buf) == 0
w = np.array(buf[:], dtype=np.float64).reshape(1024, 5)[:nwg]
w = (w - w[:, 0].min()) / 100.0
q = lambda x: "min %5.1f  median %5.1f  max %5.1f" % (x.min(), np.median(x), x.max())
print("%s launch, K = %d, %d workgroups (us since the first one entered)" % (mode, K, nwg))
print("  entry                        %s" % q(w[:, 0]))
print("  LDS cleared                  %s  (duration)" % q(w[:, 1] - w[:, 0]))
print("  first stage in LDS           %s  (duration)" % q(w[:, 2] - w[:, 1]))
print("  stages                       %s  (duration)" % q(w[:, 3] - w[:, 2]))
print("  slab stored                  %s  (duration)" % q(w[:, 4] - w[:, 3]))
print("  end                          %s" % q(w[:, 4]))
