import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import clvae_amd
from clvae_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for (M, N, K, ta) in ((100, 1, 88, False), (100, 40, 88, False), (88, 4, 128, True), (20, 20, 64, False)):
    A = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    C = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ws = ops.Workspace(dev)
    ops.gemm(torch.as_tensor(A, device=dev), torch.as_tensor(B, device=dev), C, M, N, K, ta=ta, ws=ws, split_k=1)
    torch.cuda.synchronize()
    ref = (A.T if ta else A).astype(np.float64) @ B.astype(np.float64)
    err = np.abs(C.cpu().numpy() - ref)
    bad = np.argwhere(err > 1e-3)
    print((M, N, K, ta), 'max err', err.max(), 'bad count', len(bad), 'first bad', bad[:6].tolist())
