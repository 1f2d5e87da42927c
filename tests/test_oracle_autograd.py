"""G5: the oracle's analytic gradients vs an independent torch-CPU autograd build of the same graph."""
import numpy as np
import pytest
import torch

from oracle import clvae_oracle as O


from oracle import torch_cpu as TC


def _t(a):
    return torch.tensor(a, dtype=torch.float64, requires_grad=True)


@pytest.mark.parametrize("use_x_prev,inter", [(True, 88), (False, 88), (True, 0), (False, 0)])
def test_vae_grads_match_autograd(use_x_prev, inter):
    rng = np.random.default_rng(1)
    cfg = O.vae_config(latent_dim=4, n_classes=3, use_x_prev=use_x_prev, class_weight=0.7, intermediate_dim=inter,
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
    ls, a = TC.vae_graph(tp, cfg, X, XP, WT, EW, EZ)
    total, vae, klw = ls['total'], ls['vae'], ls['kl_w']
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
    ls, a = TC.vrnn_graph(tp, cfg, tX, tXp, tW, teW, teZ)
    total = ls['total']
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


def test_torch_adam_wn_tracks_the_numpy_oracle():
    """The torch Adam-WN of the CPU baseline == oracle.adam_wn_step over three steps (fp64)."""
    rng = np.random.default_rng(4)
    p = {'a/kernel': rng.standard_normal((7, 5)), 'a/bias': rng.standard_normal(5), 'b/kernel': rng.standard_normal((5, 3))}
    tp = {k: _t(v.copy()) for k, v in p.items()}
    st, opt = O.adam_wn_init(p), TC.AdamWN(tp)
    for _ in range(3):
        g = {k: rng.standard_normal(v.shape) for k, v in p.items()}
        O.adam_wn_step(p, g, st)
        for k in tp:
            tp[k].grad = torch.tensor(g[k])
        opt.step(tp)
    for k in p:
        np.testing.assert_allclose(tp[k].detach().numpy(), p[k], rtol=1e-10, atol=1e-12, err_msg=k)


def test_cpu_baseline_timer_runs():
    cfg = O.vrnn_config(latent_dim=2, seq_length=4, n_classes=3, use_x_prev=True)
    r = TC.time_training_steps('cl_vrnn', cfg, B=4, T=4, seconds=0.5, min_steps=2, warmup=1)
    assert r['steps'] >= 2 and r['timesteps_per_s'] > 0 and r['threads'] >= 1
    cfg = O.vae_config(latent_dim=2, n_classes=3, use_x_prev=True)
    assert TC.time_training_steps('cl_vae', cfg, B=8, T=1, seconds=0.5, min_steps=2, warmup=1)['steps'] >= 2


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
