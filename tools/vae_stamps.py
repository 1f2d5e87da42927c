"""Stage timeline of the fused cl_vae step (csrc/vae_fused.hip built with -DCLV_VAE_STAMPS): workgroup 0's wall clock at
every stage boundary, median over a few launches.
  bash tools/build_variant.sh stamps "-DCLV_VAE_STAMPS" vae_fused.hip
  CLV_LIB=$PWD/abtest/stamps/libclvae_hip.so python tools/vae_stamps.py [staged]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from clvae_amd import _lib  # noqa: E402

# stamp slots: 0 start, 1 inputs staged, 2 + i after stage i, 14 + j after the per-row hook that follows stage 2j + 1
ORDER = [(0, 'start'), (1, 'zero LDS + stage inputs'), (2, 'h_w'), (3, 'wargs'), (14, 'w ~ logistic-normal + label loss'),
         (4, 'h'), (5, 'zargs'), (15, 'z + KL'), (6, 'h_dec'), (7, 'logits'), (16, 'NLL + dlogits'),
         (8, 'bwd x_decoded_mean'), (9, 'bwd decoder_h'), (17, 'gauss bwd'), (10, 'bwd zargs'), (11, 'bwd h'),
         (18, 'label bwd'), (12, 'bwd wargs'), (13, 'bwd h_w'), (19, 'end')]

dev = torch.device('cuda:0')
w = bench.WORKLOADS['cfg2']
eng, cfg = bench.make_engine(w, dev)
B = w['B']
rng = np.random.default_rng(0)
x = torch.as_tensor((rng.random((B, 88)) < 0.0443).astype(np.float32), device=dev)
xp = torch.as_tensor((rng.random((B, 88)) < 0.0443).astype(np.float32), device=dev)
oh = torch.as_tensor(np.eye(w['C'], dtype=np.float32)[rng.integers(0, w['C'], B)], device=dev)
ew = torch.randn(B, w['C'] - 1, device=dev)
ez = torch.randn(B, w['L'], device=dev)
lib = _lib.lib()
fn = lib.clv_debug_vae_stamps
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
rows = []
raw = []
STAGED = len(sys.argv) > 1 and sys.argv[1] == 'staged'      # the training step's own path: the launch assembles its byte rows
if STAGED:
    from clvae_amd.trainer import TrainStep
    X_all, Xp_all, w_all = bench.synthetic_windows(w, 4 * B, 1234, dev)
    ts = TrainStep(eng, seed=1234, use_graph=False)
    ts.bind_batches(X_all, Xp_all, w_all, idx=None, period=4, stride=B)
    print("staged path (TrainStep.bind_batches, eager)")
for it in range(12):
    if STAGED:
        ts.step()
    else:
        eng.loss_and_grads(x, xp, oh, ew, ez)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert fn(buf, 64) == 0
    rows.append(np.array([buf[i] for i, _ in ORDER], dtype=np.float64))
    raw.append(list(buf))
d = np.diff(np.array(rows[2:]), axis=1) * 10.0 / 1000.0      # 100 MHz ticks -> us
med = np.median(d, axis=0)
for (_, n), v in zip(ORDER[1:], med):
    print("%-36s %6.2f us" % (n, v))
print("%-36s %6.2f us" % ("total (workgroup 0)", med.sum()))
# inside four stages (wave 0): top -> weights requested -> product done -> weight gradients issued -> barrier passed
for k, (name, end_slot) in enumerate([('h', 4), ('zargs', 5), ('bwd x_decoded_mean', 8), ('bwd decoder_h', 9)]):
    sl = [20 + 4 * k + i for i in range(4)] + [end_slot]
    t = np.median(np.array([[r_[i] for i in sl] for r_ in raw[2:]], dtype=np.float64), axis=0)
    dd = np.diff(t) * 10.0 / 1000.0
    print("  %-20s request weights %5.2f | wait + product %5.2f | weight gradients %5.2f | barrier %5.2f us" % ((name,) + tuple(dd)))
