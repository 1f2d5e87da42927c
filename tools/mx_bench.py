"""Times the large-batch LSTM kernels (csrc/lstm_mx.hip) against the launches they replace (sparse projection + sequence
forward, sequence backward) on synthetic piano-roll frames.  python tools/mx_bench.py [B] [T] [L]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clvae_amd  # noqa: F401,E402
from clvae_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = int(sys.argv[3]) if len(sys.argv) > 3 else 32
H, D = 88, 88
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
ld = (D + L + 3) // 4 * 4
XZ = np.zeros((B * T, ld), np.float32)
XZ[:, :D] = rng.random((B * T, D)) < 0.0443
XZ[:, D:D + L] = rng.standard_normal((B * T, L))
XZd = t(XZ)
K = t(rng.standard_normal((D + L, 4 * H)) * 0.2)
U = t(rng.standard_normal((H, 4 * H)) * 0.1)
rb = t(rng.standard_normal((B, 4 * H)) * 0.3)
dhs = t(rng.standard_normal((B * T, H)) * 0.1)
hs = torch.empty(B * T, H, device=dev)
cs = torch.empty(B * T, 2 * H, device=dev)
gates = torch.empty(B * T, 4 * H, device=dev)
dzsum = torch.empty(B, 4 * H, device=dev)
dZ = torch.empty(B * T, L, device=dev)
Kz = K[D:]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def old_fwd(z):
    ops.sparse_proj(B * T, D, 4 * H, XZd, ld, K, gates)
    if z:      # (until round 6 the 4x4x1 f32-MFMA forward added z_t . Kz in the kernel; now: one more row gather / GEMM)
        ops.gemm(XZd[:, D:], Kz, gates, B * T, 4 * H, L, lda=ld, beta=1.0, ws=ops.Workspace(dev))
    ops.lstm_seq_fwd(B, T, gates, rb, U, hs, cs, gates)


def new_fwd(z):
    ops.lstm_mx_fwd(B, T, XZd, ld, D, K, XZd[:, D:] if z else None, ld, L if z else 0, Kz if z else None, rb, U, hs, gates, cs)


res = {}
for z in (0, 1):
    if B >= 768:
        res['old_fwd_z%d' % z] = timeit(lambda: old_fwd(z))
        old_fwd(z)
        g0 = gates.clone()
        res['old_bwd_z%d' % z] = timeit(lambda: (gates.copy_(g0), ops.lstm_seq_bwd_z(B, T, U, dhs, cs, gates, dzsum, Kz, L, dZ, L)
                                                 if z else ops.lstm_seq_bwd(B, T, U, dhs, cs, gates, dzsum)))
        res['copy'] = timeit(lambda: gates.copy_(g0))
    res['new_fwd_z%d' % z] = timeit(lambda: new_fwd(z))
    new_fwd(z)
    g1 = gates.clone()
    res['new_bwd_z%d' % z] = timeit(lambda: (gates.copy_(g1), ops.lstm_mx_bwd(B, T, U, dhs, cs, gates, dzsum, Kz if z else None,
                                                                              L if z else 0, dZ if z else None, L)))
    res['copy'] = timeit(lambda: gates.copy_(g1))
print("B %d T %d L %d (us per launch; bwd rows include one %d-us copy of the gate buffer)" % (B, T, L, round(res['copy'])))
for k in sorted(res):
    print("  %-12s %9.1f" % (k, res[k]))
