"""Time clv_latent_head_fwd / _bwd at the config-5 size (R = 262144 encoder states, latent 32) and check the variant's
outputs against the in-tree build's on the same inputs.
Usage (GPU box): python tools/latent_bench.py [path/to/other/libclvae_hip.so]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import _lib, ops

if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(3)
f = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


R, H, L = int(os.environ.get('R', 262144)), 88, int(os.environ.get('L', 32))
ws = ops.Workspace(dev)
hs, Wz, bz, eps = torch.tanh(f(R, H)), f(H, 2 * L) * 0.2, f(2 * L) * 0.1, f(R, L)
zargs, Z, rowkl = torch.empty(R, 2 * L, device=dev), torch.empty(R, L, device=dev), torch.empty(R, device=dev)
dZ, dhs, dWz, dbz = f(R, L), torch.empty(R, H, device=dev), torch.empty(H, 2 * L, device=dev), torch.empty(2 * L, device=dev)
nz = ops.noise_draw(7, 1, 0) if os.environ.get('NOISE') == '1' else None      # the training step's form: eps drawn in the kernel
tf = timeit(lambda: ops.latent_head_fwd(R, H, L, hs, Wz, bz, eps, zargs, Z, L, rowkl, noise=nz))
tb = timeit(lambda: ops.latent_head_bwd(R, H, L, hs, Wz, zargs, eps, dZ, L, 1.0 / R, dhs, dWz, dbz, ws))
torch.cuda.synchronize()
sums = [float(t.double().abs().sum()) for t in (zargs, Z, rowkl, dhs, dWz, dbz)]
print("%s: latent head R=%d L=%d  fwd %.1f us  bwd (+ its reduce) %.1f us   |sums| %s" % (
    os.path.basename(os.path.dirname(_lib.LIB_PATH)) or '.', R, L, tf, tb, " ".join("%.9e" % v for v in sums)))
