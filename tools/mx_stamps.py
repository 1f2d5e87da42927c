"""Where a step of the large-batch LSTM forward goes (csrc/lstm_mx.hip built with -DMX_STAMPS): shader-clock stamps of every
wave of workgroup 0 at steps 64..71 of a 1024 x 256 launch.
  bash tools/build_variant.sh mxstamps "-DMX_STAMPS -fno-slp-vectorize" lstm_mx.hip
  CLV_LIB=$PWD/abtest/mxstamps/libclvae_hip.so python tools/mx_stamps.py [z]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd import _lib, ops  # noqa: E402

withz = 'z' in sys.argv[1:]
bwd = 'bwd' in sys.argv[1:]
B, T, L, H, D = 1024, 256, 32, 88, 88
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
ld = 120
XZ = np.zeros((B * T, ld), np.float32)
XZ[:, :D] = rng.random((B * T, D)) < 0.0443
XZ[:, D:D + L] = rng.standard_normal((B * T, L))
XZd = t(XZ)
K, U, rb = t(rng.standard_normal((D + L, 4 * H)) * 0.2), t(rng.standard_normal((H, 4 * H)) * 0.1), t(rng.standard_normal((B, 4 * H)) * 0.3)
hs, cs, gates = torch.empty(B * T, H, device=dev), torch.empty(B * T, 2 * H, device=dev), torch.empty(B * T, 4 * H, device=dev)
fn = _lib.lib().clv_debug_mx_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
fp = _lib.lib().clv_debug_mx_pstamps
fp.restype = ctypes.c_int
fp.argtypes = [ctypes.c_void_p]
rows, prow = [], []
dhs = t(rng.standard_normal((B * T, H)) * 0.1)
dzsum, dZ = torch.empty(B, 4 * H, device=dev), torch.empty(B * T, L, device=dev)
for it in range(6):
    ops.lstm_mx_fwd(B, T, XZd, ld, D, K, XZd[:, D:] if withz else None, ld, L if withz else 0, K[D:] if withz else None, rb, U,
                    hs, gates, cs)
    if bwd:
        ops.lstm_mx_bwd(B, T, U, dhs, cs, gates, dzsum, K[D:] if withz else None, L if withz else 0, dZ if withz else None, L)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 512)()
    assert fn(buf) == 0
    rows.append(np.array(buf[:], dtype=np.float64).reshape(8, 8, 8)[:, :, :7])
    b2 = (ctypes.c_ulonglong * 32)()
    assert fp(b2) == 0
    prow.append(np.array(b2[:], dtype=np.int64).astype(np.float64).reshape(8, 4))
a = np.array(rows[2:])                      # [launch, step, wave, stamp]
if bwd:
    a = a[..., :4]
    print("backward kernel, cycles (median); unit waves 0..5 (+ latent waves record nothing)")
    print("%-40s" % "wave" + "".join("%8d" % w for w in range(6)))
    for k, nm in enumerate(['top -> dh_rec (11 reads, 33 MFMAs, butterfly)', 'cell math, dz pieces, stores', 'loads issued -> at the barrier']):
        print("%-40s" % nm[:40] + "".join("%8.0f" % np.median(a[:, :, w, k + 1] - a[:, :, w, k]) for w in range(6)))
    print("%-40s" % "whole step" + "".join("%8.0f" % np.median(np.diff(a[:, :, w, 0], axis=1)) for w in range(6)))
    last = a[:, :, :6, 3].max(axis=2, keepdims=True)
    print("%-40s" % "waits at the barrier" + "".join("%8.0f" % np.median(last[:, :, 0] - a[:, :, w, 3]) for w in range(6)))
    sys.exit(0)
names = ['top -> B operands in registers', 'B operands -> last MFMA result', 'MFMA -> gather FMAs done (wave 7: + compaction)',
         'gather -> butterfly done', 'gate math', 'stores + LDS writes issued -> at the barrier']
d = np.diff(a, axis=3)
print("cycles (median over %d launches x 8 steps); step = top of step t+1 - top of step t" % a.shape[0])
step = np.median(np.diff(a[:, :, :, 0], axis=1), axis=(0, 1))
last = a[..., 6].max(axis=2, keepdims=True)
early = np.median((last - a[..., 6]), axis=(0, 1))
print("%-52s" % "wave" + "".join("%8d" % w for w in range(8)))
for k, nm in enumerate(names):
    print("%-52s" % nm + "".join("%8.0f" % np.median(d[:, :, w, k]) for w in range(8)))
print("%-52s" % "top -> barrier" + "".join("%8.0f" % np.median(a[:, :, w, 6] - a[:, :, w, 0]) for w in range(8)))
print("%-52s" % "waits at the barrier for the last wave" + "".join("%8.0f" % early[w] for w in range(8)))
print("%-52s" % "whole step" + "".join("%8.0f" % step[w] for w in range(8)))
pp = np.median(np.array(prow[2:]).reshape(-1, 4), axis=0)
print("producer wave, inside the compaction: MFMA result -> frame values in registers %.0f, -> prefix + row count %.0f, -> list "
      "written %.0f, -> out %.0f cycles" % tuple(pp))
