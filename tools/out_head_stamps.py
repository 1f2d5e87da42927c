"""Where the output-head kernel's time goes at configuration 3 (one block of 128 rows per workgroup): shader-clock stamps
of wave 0 / workgroup 0 at the phase boundaries (csrc/out_head.hip built with -DOH_STAMPS).
  bash tools/build_variant.sh ohstamps "-DOH_STAMPS" out_head.hip
  CLV_LIB=$PWD/abtest/ohstamps/libclvae_hip.so python tools/out_head_stamps.py"""
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
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'cfg3']
eng, cfg = bench.make_engine(w, dev)
X, Xp, wv = bench.synthetic_windows(w, w['B'], 7, dev)
ts = TrainStep(eng, seed=1, use_graph=False)
ib = torch.arange(w['B'], device=dev)
fn = _lib.lib().clv_debug_out_head_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
rows = []
for it in range(6):
    ts.gather_batch(X, Xp, wv, ib)
    ts.step()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert fn(buf) == 0
    rows.append(np.array(buf[:9], dtype=np.float64))
d = np.median(np.diff(np.array(rows[2:]), axis=1), axis=0)
names = ['bias / first loads issued, Wo -> LDS, barrier', 'hs rows -> LDS tile (waits for the global loads)',
         'logits = hs.Wo (22 k-steps x 6 MFMA)', 'Bernoulli NLL, dl -> LDS tile, stores', 'dhs = dl.Wo^T + stores',
         'barrier', 'weight gradient (32 k-steps x 5 MFMA)', 'slab store']
for n, v in zip(names, d):
    print("%-52s %7.0f cycles  %5.2f us" % (n, v, v / 2340.0))
print("%-52s %7.0f cycles  %5.2f us (last block of the workgroup)" % ("sum", d.sum(), d.sum() / 2340.0))
