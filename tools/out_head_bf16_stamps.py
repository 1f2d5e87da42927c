"""Where out_head_bf16_kernel's time goes: shader-clock stamps of wave 0 / workgroup 0 at the phase boundaries of its LAST
block (csrc/out_head_bf16.hip built with -DOB_STAMPS).
  bash tools/build_variant.sh obstamps "-DOB_STAMPS -fno-slp-vectorize" out_head_bf16.hip
  CLV_LIB=$PWD/abtest/obstamps/libclvae_hip.so R=32768 python tools/out_head_bf16_stamps.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import clvae_amd  # noqa: E402,F401
from clvae_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
R, H = int(os.environ.get('R', 32768)), 88
f = lambda *s: torch.randn(*s, device=dev)
ws = ops.Workspace(dev)
hs, Wo, bo = torch.tanh(f(R, H)), f(H, H) * 0.3, f(H)
Y = (torch.rand(R, H, device=dev) < 0.05).float()
logits, rn, dhs, dWo, dbo = f(R, H), f(R), f(R, H), f(H, H), f(H)
fn = _lib.lib().clv_debug_out_head_bf16_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
rows = []
for it in range(8):
    ops.out_head_train(R, H, H, hs, Wo, bo, Y, 1.0 / R, rn, dhs, dWo, dbo, ws, logits=logits)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert fn(buf) == 0
    rows.append(np.array(buf[:11], dtype=np.float64))
full = np.array([np.array(r) for r in rows[2:]])
rows = full[:, :8]
d = np.median(np.diff(rows, axis=1), axis=0)
names = ['Wo / bo -> two LDS images of bf16 pieces, barrier', '(block loop top) hs row loads issued',
         'P1 logits (waits for hs; 108 MFMA 16x16x32), y requested', 'Bernoulli NLL, stores, hs^T requested, dl -> LDS tile',
         'P3 weight gradient (54 MFMA 32x32x16)', 'P2 dhs (108 MFMA 16x16x32) + stores',
         'eight waves\' gradients through LDS, slab store']
pro = np.median(full[:, [8, 9, 10, 1]] - full[:, [0]], axis=0)
print("prologue: loads issued at %.0f, arrived %.0f, images stored %.0f, past the barrier %.0f cycles" % tuple(pro))
blocks = (R + 127) // 128
per_wg = (blocks + 255) // 256
for i, (n, v) in enumerate(zip(names, d)):
    note = '  (spans the earlier blocks too)' if i == 1 and per_wg > 1 else ''
    print("%-58s %7.0f cycles  %5.2f us%s" % (n, v, v / 2340.0, note))
print("R = %d: %d blocks per workgroup; cycles of the shader clock (2.34 GHz)" % (R, per_wg))

fw = _lib.lib().clv_debug_out_head_bf16_wg
fw.restype = ctypes.c_int
fw.argtypes = [ctypes.c_void_p]
wb = (ctypes.c_ulonglong * 1024)()
assert fw(wb) == 0
w = np.array(wb[:], dtype=np.float64).reshape(256, 4)[:min(256, blocks)]
t0 = w[:, 0].min()
w = (w - t0) / 100.0       # us
q = lambda x: "min %.1f  median %.1f  max %.1f" % (x.min(), np.median(x), x.max())
print("per workgroup (us since the first one started): start    %s" % q(w[:, 0]))
print("                                                images   %s  (duration)" % q(w[:, 1] - w[:, 0]))
print("                                                row loop %s  (duration, wave 0)" % q(w[:, 2] - w[:, 1]))
print("                                                combine  %s  (duration, incl. waiting for the other waves)" % q(w[:, 3] - w[:, 2]))
print("                                                end      %s" % q(w[:, 3]))
