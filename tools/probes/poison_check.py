import torch, sys
sys.path.insert(0, '/root/repo/tests')
nan = float('nan')
blocks = [torch.full((64 << 20,), nan, dtype=torch.float32, device='cuda') for _ in range(3)]
blocks += [torch.full((n,), nan, dtype=torch.float32, device='cuda') for n in (1 << 22, 1 << 20, 1 << 18) for _ in range(8)]
blocks += [torch.full((100000,), nan, dtype=torch.float32, device='cuda') for _ in range(64)]
blocks += [torch.full((n,), nan, dtype=torch.float32, device='cuda') for n in (16384, 2048, 256, 16) for _ in range(256)]
torch.cuda.synchronize(); del blocks
for n in (10, 1000, 12880, 200000, 3000000, 40000000):
    t = torch.empty(n, dtype=torch.float32, device='cuda')
    print(n, 'nan fraction', float(torch.isnan(t).float().mean()))
