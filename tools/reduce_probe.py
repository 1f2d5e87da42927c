"""Which pending reduction makes the multi-reduce launch as long as it is at configuration 3?  Queues the slabs of the
LSTM kernel gradients (one pair launch or two single launches) and of the output head, and times flush() over subsets.
Usage (GPU box): python tools/reduce_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import ops

dev = torch.device('cuda:0')
K, T, N, nx, nh = 32768, 128, 352, 88, 88
f = lambda *s: torch.randn(*s, device=dev)
X = (torch.rand(K, 88, device=dev) < 0.0443).float()
XZ = torch.zeros(K, 92, device=dev); XZ[:, :88] = X; XZ[:, 88:90] = f(K, 2)
hs, dz = torch.tanh(f(K, nh)), f(K, N)
ge, gue = f(nx, N), f(nh, N)
gd, gud = f(nx + 2, N), f(nh, N)
Wo, bo, Y = f(88, 88) * 0.3, f(88), X
rn, dhs, dWo, dbo = f(K), f(K, 88), f(88, 88), f(88)
ws = ops.Workspace(dev)
pe = (K, N, X, 88, nx, True, hs, nh, nh, T, None, 88, 0, dz, ge, gue, None)
pd = (K, N, XZ, 92, nx, True, hs, nh, nh, T, XZ[:, 88:], 92, 2, dz, gd[:88], gud, gd[88:])


def run(kind):
    rq = ops.ReduceQueue(dev)
    if 'pair' in kind:
        ops.lstm_wgrad_pair(pd, pe, None, defer=rq)
    if 'two' in kind:
        ops.lstm_wgrad(*pd, ws, defer=rq)
        ops.lstm_wgrad(*pe, ws, defer=rq)
    if 'head' in kind:
        ops.out_head_train(K, 88, 88, hs, Wo, bo, Y, 1.0 / K, rn, dhs, dWo, dbo, ws, defer=rq)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(30):
        n = rq.n
        a.record()
        rq.flush()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
        rq.n = n            # the slabs are still there: reduce them again
    ts.sort()
    return ts[len(ts) // 2]


for kind in ('pair', 'two', 'head', 'pair+head', 'two+head'):
    print("%-10s flush %.1f us" % (kind, run(kind)))
