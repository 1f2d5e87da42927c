"""Where a step of the pair forward's decoder chain goes (csrc/lstm_pair.hip built with -DPAIR_STAMPS): shader-clock
stamps of the first decoder wave of workgroup 0 at steps 32..39 of a config-3 launch.
  bash tools/build_variant.sh pstamps "-DPAIR_STAMPS" lstm_pair.hip
  CLV_LIB=$PWD/abtest/pstamps/libclvae_hip.so python tools/pair_stamps.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from clvae_amd import _lib  # noqa: E402
from clvae_amd.trainer import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
w = bench.WORKLOADS['cfg3']
eng, cfg = bench.make_engine(w, dev)
X, Xp, wv = bench.synthetic_windows(w, w['B'], 7, dev)
ts = TrainStep(eng, seed=1, use_graph=False)
ib = torch.arange(w['B'], device=dev)
fn = _lib.lib().clv_debug_pair_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
fa = _lib.lib().clv_debug_pair_arrive
fa.restype = ctypes.c_int
fa.argtypes = [ctypes.c_void_p]
rows, arr = [], []
for it in range(6):
    ts.gather_batch(X, Xp, wv, ib)
    ts.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert fn(buf) == 0
    rows.append(np.array(buf[:], dtype=np.float64).reshape(8, 8))
    b2 = (ctypes.c_ulonglong * 192)()
    assert fa(b2) == 0
    arr.append(np.array(b2[:], dtype=np.float64).reshape(8, 12, 2))
a = np.array(rows[2:])                       # [launch, step, stamp]
d = np.diff(a, axis=2)                       # between consecutive stamps of a step
names = ['top -> first h bytes', 'first -> last h bytes', 'h . U (44 packed FMAs)', 'slice reduce', 'cell (gates, c, h)',
         'LDS write, selects, stores issued', 'barrier']
med = np.median(d.reshape(-1, 7), axis=0)
for n, v in zip(names, med):
    print("%-36s %7.0f cycles" % (n, v))
step = np.median(np.diff(a[:, :, 0], axis=1))
print("%-36s %7.0f cycles (stamp 0 of consecutive steps)" % ("whole step", step))

# per wave: cycles from the top of its step to its arrival at the barrier, and how long before the LAST arrival it got there
A = np.array(arr[2:])                        # [launch, step, wave, (top, arrive)]
work = np.median((A[..., 1] - A[..., 0]).reshape(-1, 12), axis=0)
last = A[..., 1].max(axis=2, keepdims=True)
early = np.median((last - A[..., 1]).reshape(-1, 12), axis=0)
for w_ in range(12):
    print("wave %2d (%s %d, SIMD %d): top -> barrier %5.0f cycles, waits %5.0f for the last wave" %
          (w_, 'enc' if w_ < 6 else 'dec', w_ % 6, w_ % 4, work[w_], early[w_]))
