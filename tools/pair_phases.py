"""Where the waves of a pair-forward workgroup spend a step (csrc/lstm_pair.hip built with -DPAIR_PHASES): per-wave sums of
shader-clock intervals over all steps of a config-3 launch, accumulated in SGPRs (no stores inside the loop).
  bash tools/build_variant.sh pphases "-DPAIR_PHASES -fno-slp-vectorize" lstm_pair.hip
  CLV_LIB=$PWD/abtest/pphases/libclvae_hip.so python tools/pair_phases.py"""
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
fn = _lib.lib().clv_debug_pair_phases
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
runs = []
for it in range(6):
    ts.gather_batch(X, Xp, wv, ib)
    ts.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_uint * 48)()
    assert fn(buf) == 0
    runs.append(np.array(buf[:], dtype=np.float64).reshape(12, 4))
a = np.median(np.array(runs[2:]), axis=0) / w['T']          # cycles per step
print("cycles per step (median of 4 launches, T = %d)" % w['T'])
print("%-22s %10s %12s %12s %10s %8s" % ("wave", "input wait", "reads+FMAs", "cell+stores", "barrier", "sum"))
for i in range(12):
    print("%-22s %10.0f %12.0f %12.0f %10.0f %8.0f" % ("%s %d (SIMD %d)" % ('enc' if i < 6 else 'dec', i % 6, i % 4),
                                                      a[i, 0], a[i, 1], a[i, 2], a[i, 3], a[i].sum()))
