"""G5: the oracle's analytic gradients vs an independent torch-CPU autograd build of the same graph."""
import numpy as np
import pytest
import torch

from oracle import clvae_oracle as O


def _t(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


def _hs(z):
    return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


def _bce(a, y):
    l = torch.clamp(a, -O.LOGIT_CLIP_HI, O.LOGIT_CLIP_HI)
    return (torch.clamp(l, min=0) - l * y + torch.log1p(torch.exp(-l.abs()))).sum(-1)


def _cce(w, y, scale):
    q = w + O.W2_SHIFT
    n = q / q.sum(-1, keepdim=True)
    return -scale * (y * torch.log(torch.clamp(n, O.EPS_K, 1 - O.EPS_K))).sum(-1)


def _logitnormal(m, lv, eps):
    s = m + torch.exp(lv / 2) * eps
    s0 = torch.cat([s, torch.zeros_like(s[..., :1])], -1)
    e = torch.exp(s0)
    return e / e.sum(-1, keepdim=True)


def _lstm(xs, k, r, b, act):
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


@pytest.mark.parametrize("use_x_prev", [True, False])
def test_vae_grads_match_autograd(use_x_prev):
    rng = np.random.default_rng(1)
    cfg = O.vae_config(latent_dim=4, n_classes=3, use_x_prev=use_x_prev, class_weight=0.7,
                       kl_weight=0.3, w_kl_weight=0.9, w_log_var_prior=0.4)
    B, D, C, L = 6, cfg['D'], cfg['C'], cfg['L']
    p = O.vae_init_params(cfg, seed=3)
    for k in p:
        if k.endswith('bias'):
            p[k] = 0.1 * rng.standard_normal(p[k].shape)
    x = (rng.random((B, D)) < 0.1).astype(np.float64)
    xp = (rng.random((B, D)) < 0.1).astype(np.float64)
    wt = np.eye(C)[rng.integers(0, C, B)]
    ew, ez = rng.standard_normal((B, C - 1)), rng.standard_normal((B, L))
    out = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)

    tp = {k: _t(v) for k, v in p.items()}
    X, XP, WT, EW, EZ = map(lambda a: torch.tensor(a), (x, xp, wt, ew, ez))
    hw = torch.relu(X @ tp['h_w/kernel'] + tp['h_w/bias'])
    wm = hw @ tp['w_mean/kernel'] + tp['w_mean/bias']
    wlv = hw @ tp['w_log_var/kernel'] + tp['w_log_var/bias']
    w = _logitnormal(wm, wlv, EW)
    h = torch.relu(torch.cat([X, w], -1) @ tp['h/kernel'] + tp['h/bias'])
    zm = h @ tp['z_mean/kernel'] + tp['z_mean/bias']
    zlv = h @ tp['z_log_var/kernel'] + tp['z_log_var/bias']
    z = zm + torch.exp(zlv / 2) * EZ
    wz = torch.cat([w, XP, z], -1) if use_x_prev else torch.cat([w, z], -1)
    hd = torch.relu(wz @ tp['decoder_h/kernel'] + tp['decoder_h/bias'])
    a = hd @ tp['x_decoded_mean/kernel'] + tp['x_decoded_mean/bias']
    pr = cfg['w_log_var_prior']
    vae = _bce(a, X).mean()
    klz = (-0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv)).sum(-1)).mean()
    wrec = _cce(w, WT, C - 1).mean()
    klw = (-0.5 * (1 - pr + wlv - torch.exp(wlv) / np.exp(pr) - wm ** 2 / np.exp(pr)).sum(-1)).mean()
    total = vae + cfg['w_kl_weight'] * klw + cfg['class_weight'] * wrec + cfg['kl_weight'] * klz
    total.backward()
    assert abs(total.item() - out['total']) < 1e-10
    assert abs(vae.item() - out['vae']) < 1e-10 and abs(klw.item() - out['kl_w']) < 1e-10
    np.testing.assert_allclose(a.detach().numpy(), out['cache']['logits'], atol=1e-12)
    for k in p:
        np.testing.assert_allclose(out['grads'][k], tp[k].grad.numpy(), rtol=1e-8, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("use_x_prev,gate_act", [(True, 'hard_sigmoid'), (False, 'hard_sigmoid'), (True, 'sigmoid')])
def test_vrnn_grads_match_autograd(use_x_prev, gate_act):
    rng = np.random.default_rng(2)
    cfg = O.vrnn_config(original_dim=12, intermediate_dim=8, latent_dim=3, seq_length=5, n_classes=4,
                        use_x_prev=use_x_prev, class_weight=0.6, kl_weight=0.4, w_kl_weight=0.8,
                        w_log_var_prior=-0.3, gate_act=gate_act)
    B, T, D, C, L = 4, cfg['T'], cfg['D'], cfg['C'], cfg['L']
    p = O.vrnn_init_params(cfg, seed=5)
    for k in p:     # make the hard-sigmoid saturate sometimes
        if 'encoder_h' in k or 'decoder_h' in k:
            p[k] = p[k] * 3.0
    X = (rng.random((B, T, D)) < 0.3).astype(np.float64)
    Xp = (rng.random((B, T, D)) < 0.3).astype(np.float64)
    wt = np.eye(C)[rng.integers(0, C, B)]
    eW, eZ = rng.standard_normal((B, C - 1)), rng.standard_normal((B, T, L))
    out = O.vrnn_loss_and_grads(p, cfg, X, Xp, wt, eW, eZ)

    tp = {k: _t(v) for k, v in p.items()}
    tX, tXp, tW, teW, teZ = map(lambda a: torch.tensor(a), (X, Xp, wt, eW, eZ))
    act = _hs if gate_act == 'hard_sigmoid' else torch.sigmoid
    hW = torch.relu(tX.reshape(B, -1) @ tp['hW/kernel'] + tp['hW/bias'])
    wa = hW @ tp['Wargs/kernel'] + tp['Wargs/bias']
    wm, wlv = wa[:, :C - 1], wa[:, C - 1:]
    W = _logitnormal(wm, wlv, teW)
    Wrep = W[:, None, :].expand(B, T, C)
    eh = _lstm(torch.cat([tX, Wrep], -1), tp['encoder_h/kernel'], tp['encoder_h/recurrent_kernel'],
               tp['encoder_h/bias'], act)
    zm = eh @ tp['Z_mean/kernel'] + tp['Z_mean/bias']
    zlv = eh @ tp['Z_log_var/kernel'] + tp['Z_log_var/bias']
    Z = zm + torch.exp(zlv / 2) * teZ
    din = torch.cat([tXp, Z, Wrep], -1) if use_x_prev else torch.cat([Z, Wrep], -1)
    dh = _lstm(din, tp['decoder_h/kernel'], tp['decoder_h/recurrent_kernel'], tp['decoder_h/bias'], act)
    a = dh @ tp['X_decoded_mean/kernel'] + tp['X_decoded_mean/bias']
    pr = cfg['w_log_var_prior']
    vae = _bce(a, tX).mean()
    klz = (-0.5 * (1 + zlv - zm ** 2 - torch.exp(zlv)).sum(-1)).mean()
    wrec = _cce(W, tW, C - 1).mean()
    klw = (-0.5 * (1 - pr + wlv - torch.exp(wlv) / np.exp(pr) - wm ** 2 / np.exp(pr)).sum(-1)).mean()
    total = vae + cfg['w_kl_weight'] * klw + cfg['class_weight'] * wrec + cfg['kl_weight'] * klz
    total.backward()
    assert abs(total.item() - out['total']) < 1e-10
    np.testing.assert_allclose(a.detach().numpy(), out['cache']['logits'], atol=1e-12)
    for k in p:
        np.testing.assert_allclose(out['grads'][k], tp[k].grad.numpy(), rtol=1e-7, atol=1e-12, err_msg=k)


def test_lstm_matches_torch_nn_lstm_with_sigmoid_gates():
    """Same i,f,g,o order as torch.nn.LSTM; Keras kernel [in,4H] == weight_ih.T."""
    rng = np.random.default_rng(0)
    B, T, In, H = 3, 7, 5, 6
    k = rng.standard_normal((In, 4 * H)) * 0.5
    r = rng.standard_normal((H, 4 * H)) * 0.5
    b = rng.standard_normal(4 * H) * 0.1
    xs = rng.standard_normal((B, T, In))
    hs, _ = O.lstm_forward(xs, k, r, b, gate_act='sigmoid')
    m = torch.nn.LSTM(In, H, batch_first=True).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.tensor(k.T)); m.weight_hh_l0.copy_(torch.tensor(r.T))
        m.bias_ih_l0.copy_(torch.tensor(b)); m.bias_hh_l0.zero_()
        ref, _ = m(torch.tensor(xs))
    np.testing.assert_allclose(hs, ref.numpy(), atol=1e-12)


def test_adam_wn_first_step_keeps_weightnorm_invariants():
    rng = np.random.default_rng(0)
    p = {'a/kernel': rng.standard_normal((5, 4)), 'a/bias': rng.standard_normal(4)}
    g = {k: rng.standard_normal(v.shape) for k, v in p.items()}
    p0 = {k: v.copy() for k, v in p.items()}
    st = O.adam_wn_init(p)
    O.adam_wn_step(p, g, st)
    # after the update W = s*V' and s = g'/||V'||  =>  column norms of W equal the new g parameter
    V = p['a/kernel'] / st['s']['a/kernel']
    np.testing.assert_allclose(np.sqrt((p['a/kernel'] ** 2).sum(0)),
                               st['s']['a/kernel'] * np.sqrt((V ** 2).sum(0)), rtol=1e-12)
    # bias: plain Adam, first step moves by lr * sign(g) (up to eps)
    np.testing.assert_allclose(p['a/bias'], p0['a/bias'] - 1e-3 * np.sign(g['a/bias']), atol=1e-9)
    # plain adam switch
    q = {k: v.copy() for k, v in p0.items()}
    st2 = O.adam_wn_init(q, weightnorm=False)
    O.adam_wn_step(q, g, st2)
    np.testing.assert_allclose(q['a/kernel'], p0['a/kernel'] - 1e-3 * np.sign(g['a/kernel']), atol=1e-9)
