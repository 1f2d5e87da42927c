"""Where and when the workgroups of one vrnn_front_kernel launch ran (csrc/label_head.hip built with -DFRONT_STAMPS):
  bash tools/build_variant.sh frontstamps "-DFRONT_STAMPS" label_head.hip
  CLV_LIB=$PWD/abtest/frontstamps/libclvae_hip.so python tools/front_timeline.py
Runs the configuration-3 step eagerly a few times and prints, for the last front launch: per role the start / end times, and how
the two roles shared the CUs (HW_ID: CU, SH, SE; XCC_ID)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd import _lib  # noqa: E402
from clvae_amd.engine import VrnnEngine  # noqa: E402
from clvae_amd.trainer import TrainStep  # noqa: E402
from oracle import clvae_oracle as O  # noqa: E402

dev = torch.device('cuda:0')
B, T = 256, 128
cfg = O.vrnn_config(latent_dim=2, seq_length=T, n_classes=10, use_x_prev=True)
eng = VrnnEngine(cfg, B, dev)
eng.P.set_weights({k: np.asarray(v, np.float32) for k, v in O.vrnn_init_params(cfg, seed=1).items()})
rng = np.random.default_rng(0)
n = 2 * B
win = rng.random((n, T + 1, 88)) < 0.0443
u8 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)
keys = torch.as_tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, n)], device=dev)
ts = TrainStep(eng, seed=1, use_graph=False)
ts.bind_batches(u8(win[:, 1:].reshape(n, -1)), u8(win[:, :-1].reshape(n, -1)), keys, idx=None, period=2, stride=B)
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
fn = _lib.lib().clv_debug_front_wg
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * (2048 * 6))()
assert fn(buf) == 0
w = np.array(buf[:], dtype=np.uint64).reshape(2048, 6)
nwg = int((w[:, 0] != 0).sum())
w = w[:nwg]
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0).astype(np.float64) / 100.0, (w[:, 1] - t0).astype(np.float64) / 100.0
hw, xcc = w[:, 2].astype(np.int64), w[:, 3].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
place = xcc * 1000 + se * 100 + sh * 10 + cu            # one number per CU
role = np.where(np.arange(nwg) < B, 0, 1)
q = lambda x: "min %5.1f  median %5.1f  max %5.1f" % (x.min(), np.median(x), x.max())
print("%d workgroups (%d label rows + %d projection), us since the first one started" % (nwg, B, nwg - B))
for r, name in ((0, 'label'), (1, 'projection')):
    m = role == r
    print("  %-11s start %s | end %s | duration %s" % (name, q(st[m]), q(en[m]), q(en[m] - st[m])))
m = role == 1
print("  projection: kernel half + row addresses in LDS after %s us" % q((w[m, 4] - w[m, 0]).astype(np.float64) / 100.0))
cus = sorted(set(place.tolist()))
mix = {}
for c in cus:
    m = place == c
    key = (int((role[m] == 0).sum()), int((role[m] == 1).sum()))
    mix[key] = mix.get(key, 0) + 1
print("  distinct CUs used: %d; (label rows, projection workgroups) per CU -> number of CUs: %s" % (len(cus), sorted(mix.items())))
# overlap in time on the same CU
both = 0
for c in cus:
    m = np.nonzero(place == c)[0]
    for i in m:
        for j in m:
            if role[i] == 0 and role[j] == 1 and st[j] < en[i] - 2 and st[i] < en[j] - 2:
                both += 1
print("  (label, projection) pairs that ran on the same CU at the same time: %d" % both)
