"""Per-k-tile phase stamps (shader clock) of the LSTM weight-gradient GEMM, one wave per k-group of one workgroup.
Needs a library built with the stamp code compiled in, e.g. in a scratch copy of the tree:
    make -C classifying-vae-lstm_amd/csrc EXTRA=-DCLV_GEMM_STAMPS
then on the GPU box: python tools/gemm_stamps.py path/to/that/libclvae_hip.so"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import clvae_amd  # noqa: F401
from clvae_amd import _lib, ops

_lib.LIB_PATH = sys.argv[1]
dev = torch.device('cuda:0')
f = lambda *s: torch.randn(*s, device=dev)
BT, H = 32768, 88
ws = ops.Workspace(dev)
dz, hs, X = f(BT, 352), f(BT, H), f(BT, 92)
g1, g2 = f(90, 352), f(H, 352)
run = lambda: ops.gemm_grouped_tn([dict(A=X, lda=92, M=90, C=g1), dict(A=hs, lda=H, M=H, C=g2, shift=1, zero_period=128)],
                                  352, BT, dz, ws)
for _ in range(5):
    run()
torch.cuda.synchronize()
h = C.CDLL(sys.argv[1])
buf = np.zeros((4, 16, 5), dtype=np.uint64)
h.clv_dbg_gemm_stamps(buf.ctypes.data_as(C.c_void_p))
t0 = buf[:, 0, 0].min()
print("columns: loop top, loads issued, MFMAs issued, LDS stores done, after barrier   (cycles since first stamp)")
for kg in range(4):
    for kt in range(16):
        print("kg%d kt%2d " % (kg, kt) + " ".join("%7d" % int(v - t0) for v in buf[kg, kt]),
              "| mfma %5d store %5d barrier %5d" % (int(buf[kg, kt, 2] - buf[kg, kt, 1]), int(buf[kg, kt, 3] - buf[kg, kt, 2]),
                                                  int(buf[kg, kt, 4] - buf[kg, kt, 3])))
