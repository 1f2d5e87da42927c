"""Do two independent LSTM sequence kernels (6 waves per row each) overlap when they share the CUs?
Times clv_lstm_seq_fwd for 256 rows alone, twice back to back on one stream, and as two launches on two streams.
Usage (GPU box): python tools/corun_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import clvae_amd  # noqa: F401
from clvae_amd import ops

dev = torch.device('cuda:0')
B, T, H = 256, 128, 88
f = lambda *s: torch.randn(*s, device=dev) * 0.3
sets = []
for i in range(2):
    sets.append(dict(xp=f(B * T, 4 * H), U=f(H, 4 * H), hs=f(B * T, H), cs=f(B * T, H), g=f(B * T, 4 * H)))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(i):
    a = sets[i]
    ops.lstm_seq_fwd(B, T, a['xp'], None, a['U'], a['hs'], a['cs'], a['g'])


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def both_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        run(0)
    with torch.cuda.stream(s2):
        run(1)
    cur.wait_stream(s1); cur.wait_stream(s2)


big = dict(xp=f(2 * B * T, 4 * H), U=f(H, 4 * H), hs=f(2 * B * T, H), cs=f(2 * B * T, H), g=f(2 * B * T, 4 * H))
print("one launch, 512 rows  %7.1f us   (two workgroups per CU if they co-reside)"
      % timed(lambda: ops.lstm_seq_fwd(2 * B, T, big['xp'], None, big['U'], big['hs'], big['cs'], big['g'])))
print("one launch            %7.1f us" % timed(lambda: run(0)))
print("two, same stream      %7.1f us" % timed(lambda: (run(0), run(1))))
print("two, two streams      %7.1f us" % timed(both_streams))
