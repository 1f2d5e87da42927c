"""Time clv_lstm_seq_fwd / _bwd at one batch size (rows per workgroup forced by CLV_LSTM_ROWS).
  CLV_LSTM_ROWS=2 python tools/lstm_rows_bench.py 512 256"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import clvae_amd  # noqa: E402,F401
from clvae_amd import ops  # noqa: E402

B, T = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda:0')
H = 88
g = torch.randn(B, T, 4 * H, device=dev)
U = torch.randn(H, 4 * H, device=dev) * 0.1
rb = torch.randn(B, 4 * H, device=dev) * 0.1
hs = torch.empty(B, T, H, device=dev); cs = torch.empty(B, T, H, device=dev)
dhs = torch.randn(B, T, H, device=dev) * 0.1
dzsum = torch.empty(B, 4 * H, device=dev)


def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


gates = g.clone()
tf = timeit(lambda: ops.lstm_seq_fwd(B, T, gates, rb, U, hs, cs, gates))
tb = timeit(lambda: ops.lstm_seq_bwd(B, T, U, dhs, cs, gates, dzsum))
print("B %5d T %4d rows/wg %s: fwd %8.1f us  bwd %8.1f us" % (B, T, os.environ.get('CLV_LSTM_ROWS', 'auto'), tf, tb))
