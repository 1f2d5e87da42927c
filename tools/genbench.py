import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import clvae_amd
from clvae_amd.engine import VrnnEngine
from clvae_amd.initializers import init_weights
dev = torch.device('cuda:0')
for N, L in ((1, 2), (256, 2), (1024, 2), (1, 32), (256, 32), (1024, 32)):
    cfg = dict(D=88, H=88, L=L, T=16, C=10, use_x_prev=True, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0, w_log_var_prior=0.0, gate_act='hard_sigmoid')
    eng = VrnnEngine(cfg, 1, dev)
    wts = init_weights(eng.P.logical, cfg, seed=0)
    wts['X_decoded_mean/bias'] = np.full_like(wts['X_decoded_mean/bias'], -3.07)   # logit(0.0443): piano-roll note density
    eng.P.set_weights(wts)
    rng = np.random.default_rng(0)
    seeds = torch.as_tensor((rng.random((N, 16, 88)) < 0.0443).astype(np.float32), device=dev)
    wv = torch.as_tensor(np.eye(10, dtype=np.float32)[rng.integers(0, 10, N)], device=dev)
    for persistent in (True, True, True, False):
        steps = 2000 if persistent else 300
        eng.generate(seeds, wv, 8, seed=1, persistent=persistent); torch.cuda.synchronize()
        t0 = time.perf_counter(); out = eng.generate(seeds, wv, steps, seed=2, persistent=persistent); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("N=%d L=%d persistent=%s: %.3f us/frame, %.0f frames/s, density %.3f" % (N, L, persistent, 1e6 * dt / (steps + 16), N * (steps + 16) / dt, float(out.mean())))
