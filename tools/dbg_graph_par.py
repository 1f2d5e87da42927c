"""Do independent branches of a captured hipGraph run concurrently?  Two latency-bound kernels (one workgroup wave
each), captured (a) back to back on one stream, (b) forked onto a side stream and joined."""
import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import clvae_amd
from clvae_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
B, nx, N = 256, 11264, 88
X = torch.as_tensor((rng.random((B, nx)) < 0.0443).astype(np.float32), device=dev)
K = torch.as_tensor(rng.standard_normal((nx, N)).astype(np.float32), device=dev)
o1, o2, o3 = (torch.zeros(B, N, device=dev) for _ in range(3))
def k(o): ops.sparse_dense(B, nx, N, X, nx, K, None, 0, o)
side = torch.cuda.Stream(device=dev)
def serial():
    k(o1); k(o2); k(o3)
def forked():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    k(o1)
    with torch.cuda.stream(side):
        k(o2)
    k(o3)
    cur.wait_stream(side)
for name, fn in (('serial', serial), ('forked', forked)):
    fn(); torch.cuda.synchronize()
    with ops.Graph() as g:
        fn()
    for _ in range(5): g.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.launch()
    torch.cuda.synchronize()
    print(name, 'graph: %.2f us per replay' % (1e6 * (time.perf_counter() - t0) / 200))
    t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize()
    print(name, 'eager: %.2f us per call' % (1e6 * (time.perf_counter() - t0) / 200))
