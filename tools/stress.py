import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import clvae_amd
from clvae_amd.engine import VrnnEngine
from clvae_amd.initializers import init_weights
from clvae_amd.trainer import TrainStep
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for (B, T, L, C, xp) in ((1024, 256, 2, 10, True), (512, 64, 8, 4, True), (300, 33, 5, 3, False), (2048, 16, 2, 10, True), (7, 200, 32, 10, True)):
    cfg = dict(D=88, H=88, L=L, T=T, C=C, use_x_prev=xp, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0, w_log_var_prior=0.0, gate_act='hard_sigmoid')
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(init_weights(eng.P.logical, cfg, seed=0))
    ts = TrainStep(eng, seed=3)
    win = torch.as_tensor((rng.random((B, T + 1, 88)) < 0.0443).astype(np.float32), device=dev)
    X, Xp = win[:, 1:].contiguous(), win[:, :-1].contiguous()
    w = torch.as_tensor(np.eye(C, dtype=np.float32)[rng.integers(0, C, B)], device=dev)
    losses = []
    for i in range(6):
        ts.stage_batch(X, Xp if xp else None, w)
        ts.step()
        losses.append(eng.losses()['total'])
    torch.cuda.synchronize()
    ok = all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]
    print((B, T, L, C, xp), 'pair' if eng.fuse_pair else 'separate', ['%.3f' % l for l in losses], 'OK' if ok else 'FAIL')
    del eng, ts
    torch.cuda.empty_cache()
