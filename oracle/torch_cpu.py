"""torch-CPU restatement of the two training graphs (TEST ORACLE / CPU baseline; see oracle/__init__.py).

An independent build of the same math as oracle/clvae_oracle.py on torch autograd: the LSTMs run as a per-timestep
loop of small matmuls and pointwise ops, the way Keras' `K.rnn` executes them (`while_loop` over T),
gradients come from autograd, and the optimizer is Adam with weight normalisation (utils/weightnorm.py:75-143).

Two users:
  * tests/test_oracle_autograd.py (G5): autograd gradients vs the numpy oracle's analytic ones, fp64;
  * bench.py `cpu_baseline`: timed on the GPU box's host cores in fp32 ("CPU restatement
    (Keras-equivalent math), not Keras").
Nothing in the product imports this module.
"""
import numpy as np
import torch

from . import clvae_oracle as O


def hard_sigmoid(z):
    return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


def bce_keras(a, y):
    """Keras binary_crossentropy summed over the notes, on logits with the epsilon clip (A.3)"""
    l = torch.clamp(a, O.LOGIT_CLIP_LO, O.LOGIT_CLIP_HI)
    return (torch.clamp(l, min=0) - l * y + torch.log1p(torch.exp(-l.abs()))).sum(-1)


def cce_keras(w, y, scale):
    q = w + O.W2_SHIFT
    n = q / q.sum(-1, keepdim=True)
    return -scale * (y * torch.log(torch.clamp(n, O.EPS_K, 1 - O.EPS_K))).sum(-1)


def logistic_normal(m, lv, eps):
    s = m + torch.exp(lv / 2) * eps
    e = torch.exp(torch.cat([s, torch.zeros_like(s[..., :1])], -1))
    return e / e.sum(-1, keepdim=True)


def lstm(xs, k, r, b, act):
    """Keras LSTM, gate blocks i, f, c, o, zero initial state, all hidden states returned (A.2)"""
    B, T, _ = xs.shape
    H = r.shape[0]
    h = torch.zeros(B, H, dtype=xs.dtype)
    c = torch.zeros(B, H, dtype=xs.dtype)
    outs = []
    for t in range(T):
        z = xs[:, t] @ k + b + h @ r
        i, f, g, o = act(z[:, :H]), act(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), act(z[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1)


def _losses(cfg, a, target, zm, zlv, w, wt, wm, wlv):
    C, pr = cfg['C'], cfg['w_log_var_prior']
    vae = bce_keras(a, target).mean()
    klz = (-0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv)).sum(-1)).mean()
    wrec = cce_keras(w, wt, C - 1).mean()
    klw = (-0.5 * (1 - pr + wlv - torch.exp(wlv) / np.exp(pr) - wm ** 2 / np.exp(pr)).sum(-1)).mean()
    total = vae + cfg['w_kl_weight'] * klw + cfg['class_weight'] * wrec + cfg['kl_weight'] * klz
    return dict(total=total, vae=vae, kl_z=klz, kl_w=klw, w_rec=wrec)


def vae_graph(tp, cfg, x, xp, wt, ew, ez):
    """cl_vae/model.py:136-219 -> (losses, logits)"""
    hw = torch.relu(x @ tp['h_w/kernel'] + tp['h_w/bias'])
    wm = hw @ tp['w_mean/kernel'] + tp['w_mean/bias']
    wlv = hw @ tp['w_log_var/kernel'] + tp['w_log_var/bias']
    w = logistic_normal(wm, wlv, ew)
    h = torch.cat([x, w], -1)
    if cfg['H'] > 0:
        h = torch.relu(h @ tp['h/kernel'] + tp['h/bias'])
    zm = h @ tp['z_mean/kernel'] + tp['z_mean/bias']
    zlv = h @ tp['z_log_var/kernel'] + tp['z_log_var/bias']
    z = zm + torch.exp(zlv / 2) * ez
    hd = torch.cat([w, xp, z], -1) if cfg['use_x_prev'] else torch.cat([w, z], -1)
    if cfg['H'] > 0:
        hd = torch.relu(hd @ tp['decoder_h/kernel'] + tp['decoder_h/bias'])
    a = hd @ tp['x_decoded_mean/kernel'] + tp['x_decoded_mean/bias']
    return _losses(cfg, a, x, zm, zlv, w, wt, wm, wlv), a


def vrnn_graph(tp, cfg, X, Xp, wt, eW, eZ):
    """cl_vrnn/model.py:169-264 -> (losses, logits)"""
    B, T, _ = X.shape
    C = cfg['C']
    act = hard_sigmoid if cfg.get('gate_act', 'hard_sigmoid') == 'hard_sigmoid' else torch.sigmoid
    hW = torch.relu(X.reshape(B, -1) @ tp['hW/kernel'] + tp['hW/bias'])
    wa = hW @ tp['Wargs/kernel'] + tp['Wargs/bias']
    wm, wlv = wa[:, :C - 1], wa[:, C - 1:]
    W = logistic_normal(wm, wlv, eW)
    Wrep = W[:, None, :].expand(B, T, C)
    eh = lstm(torch.cat([X, Wrep], -1), tp['encoder_h/kernel'], tp['encoder_h/recurrent_kernel'], tp['encoder_h/bias'], act)
    zm = eh @ tp['Z_mean/kernel'] + tp['Z_mean/bias']
    zlv = eh @ tp['Z_log_var/kernel'] + tp['Z_log_var/bias']
    Z = zm + torch.exp(zlv / 2) * eZ
    din = torch.cat([Xp, Z, Wrep], -1) if cfg['use_x_prev'] else torch.cat([Z, Wrep], -1)
    dh = lstm(din, tp['decoder_h/kernel'], tp['decoder_h/recurrent_kernel'], tp['decoder_h/bias'], act)
    a = dh @ tp['X_decoded_mean/kernel'] + tp['X_decoded_mean/bias']
    return _losses(cfg, a, X, zm, zlv, W, wt, wm, wlv), a


class AdamWN:
    """Adam with weight normalisation over a dict of tensors (utils/weightnorm.py:75-143; A.5): matrices are
    re-parameterised per output column as W = g V/||V|| with V = W/s, biases take plain Adam."""

    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        z = torch.zeros_like
        self.st = {k: dict(m=z(p), v=z(p), **(dict(s=torch.ones(p.shape[-1], dtype=p.dtype), mg=torch.zeros(p.shape[-1], dtype=p.dtype),
                                                   vg=torch.zeros(p.shape[-1], dtype=p.dtype)) if p.ndim > 1 else {}))
                   for k, p in params.items()}

    @torch.no_grad()
    def step(self, params):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, p in params.items():
            st, g = self.st[k], p.grad
            if p.ndim == 1:
                st['m'].mul_(self.b1).add_(g, alpha=1 - self.b1)
                st['v'].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                p.sub_(lr_t * st['m'] / (st['v'].sqrt() + self.eps))
            else:
                V = p / st['s']
                Vn = V.pow(2).sum(0).sqrt()
                gpar = st['s'] * Vn
                grad_g = (g * V).sum(0) / Vn
                grad_V = st['s'] * (g - (grad_g / Vn) * V)
                st['mg'].mul_(self.b1).add_(grad_g, alpha=1 - self.b1)
                st['vg'].mul_(self.b2).addcmul_(grad_g, grad_g, value=1 - self.b2)
                g_new = gpar - lr_t * st['mg'] / (st['vg'].sqrt() + self.eps)
                st['m'].mul_(self.b1).add_(grad_V, alpha=1 - self.b1)
                st['v'].mul_(self.b2).addcmul_(grad_V, grad_V, value=1 - self.b2)
                V_new = V - lr_t * st['m'] / (st['v'].sqrt() + self.eps)
                s_new = g_new / V_new.pow(2).sum(0).sqrt()
                p.copy_(V_new * s_new)
                st['s'].copy_(s_new)
            p.grad = None


def time_training_steps(model, cfg, B, T, seconds=20.0, min_steps=10, warmup=3, seed=0, density=0.0443):
    """fp32 training steps (forward, autograd backward, Adam-WN) of `model` ('cl_vrnn' | 'cl_vae') at batch B on
    this host's cores -> dict(timesteps_per_s (median step), steps, threads)."""
    import time
    rng = np.random.default_rng(seed)
    f = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
    L, C = cfg['L'], cfg['C']
    if model == 'cl_vrnn':
        init, graph = O.vrnn_init_params, vrnn_graph
        win = (rng.random((B, T + 1, 88)) < density)
        X, Xp, eZ = f(win[:, 1:]), f(win[:, :-1]), f(rng.standard_normal((B, T, L)))
    else:
        init, graph = O.vae_init_params, vae_graph
        X, Xp, eZ = f(rng.random((B, 88)) < density), f(rng.random((B, 88)) < density), f(rng.standard_normal((B, L)))
    tp = {k: f(v).requires_grad_(True) for k, v in init(cfg, seed=0, dtype=np.float32).items()}
    wt, eW = f(np.eye(C)[rng.integers(0, C, B)]), f(rng.standard_normal((B, C - 1)))
    opt = AdamWN(tp)
    times, t_start = [], time.time()
    while len(times) < warmup + min_steps or (time.time() - t_start < seconds and len(times) < warmup + 200):
        t0 = time.perf_counter()
        losses, _ = graph(tp, cfg, X, Xp, wt, eW, eZ)
        losses['total'].backward()
        opt.step(tp)
        times.append(time.perf_counter() - t0)
        if time.time() - t_start > 4 * seconds and len(times) >= warmup + 3:      # a very slow host: stop early
            break
    timed = times[warmup:]
    return dict(timesteps_per_s=B * T / float(np.median(timed)), steps=len(timed), threads=torch.get_num_threads())
