import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import clvae_amd
from clvae_amd.engine import VrnnEngine
from oracle import clvae_oracle as O
B, Tn, L, Cn = 6, 5, 2, 10
dev = torch.device('cuda:0')
def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)
def T(a): return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
res = {}
for pair in (False, True):
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True, class_weight=0.8, kl_weight=0.6, w_kl_weight=0.9, w_log_var_prior=0.2)
    cfg['fuse_pair'] = pair
    rng = np.random.default_rng(B * 1000 + Tn)
    p = {k: f32(v) for k, v in O.vrnn_init_params(cfg, seed=2).items()}
    win = (rng.random((B, Tn + 1, 88)) < 0.0443).astype(np.float64)
    X, Xp = win[:, 1:].copy(), win[:, :-1].copy()
    wt = np.eye(Cn)[rng.integers(0, Cn, B)]
    eW, eZ = f32(rng.standard_normal((B, Cn - 1))), f32(rng.standard_normal((B, Tn, L)))
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(p)
    eng.loss_and_grads(T(X), T(Xp), T(wt), T(eW), T(eZ))
    torch.cuda.synchronize()
    res[pair] = {k: getattr(eng, k).detach().cpu().numpy().copy() for k in ('dzargs', 'gates_dec', 'gates_enc', 'dzsum_dec', 'dzsum_enc', 'zargs')}
    res[pair]['grads'] = eng.P.get_weights(eng.P.grads)
for k in ('zargs', 'gates_dec', 'dzsum_dec', 'dzargs', 'gates_enc', 'dzsum_enc'):
    a, b = res[False][k], res[True][k]
    print(k, a.shape, 'maxabs ref', np.abs(a).max(), 'maxdiff', np.abs(a - b).max())
d = (res[False]['dzargs'] - res[True]['dzargs']).reshape(B, Tn, 2 * L)
print('dzargs diff per t:', np.abs(d).max(axis=(0, 2)))
print('dzargs diff per col:', np.abs(d).max(axis=(0, 1)))
print(res[False]['dzargs'].reshape(B, Tn, 2*L)[0]); print(res[True]['dzargs'].reshape(B, Tn, 2*L)[0])
for k in res[False]['grads']:
    a, b = res[False]['grads'][k], res[True]['grads'][k]
    print(k, np.abs(a - b).max() / (np.abs(a).max() + 1e-12))
