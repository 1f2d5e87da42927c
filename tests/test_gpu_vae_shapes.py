"""-m gpu: the cl_vae surface beyond 88 x 88 x 88 (round-5 verdict, "missing 3").

The reference builds cl_vae for any `--intermediate_dim`, `--intermediate_class_dim` and, through `--seq_length > 1`, for
`original_dim != 88` (cl_vae/train.py:21-30,86-96; cl_vae/model.py:130-224).  `clv_vae_fused_supported` admits every
D, H, Hc in 1..96: here the fused whole-step kernel AND the layer-by-layer chain run at widths that are not 88, not
multiples of 4, smaller than one MFMA tile, and at the edge of what the fused kernel takes; D > 96 must take the chain.
Then `cl_vae/train.py r --seq_length 2` end to end on the real JSB_Cs against an oracle loop, and the `enc_model` /
`encoder` that both `get_model`s return (cl_vae/model.py:220-224, cl_vrnn/model.py:266) against `oracle.*_forward`.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import write_jsb_cs_pickle
from oracle import clvae_oracle as O
from oracle import philox as OP

pytestmark = pytest.mark.gpu

ELBO_TOL = 1e-3        # nats per sample (the north star's tolerance)
LOGIT_TOL = 2e-4


@pytest.fixture(scope="module")
def dev():
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def T_(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def N(t):
    return t.detach().cpu().numpy().astype(np.float64)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


SHAPES = [
    # D,  H,  Hc, L,  C,  use_x_prev, B
    (88, 64, 32, 4, 2, True, 100),          # narrower hidden layers, multiples of 16
    (60, 88, 88, 4, 2, True, 48),           # fewer notes (a data set whose range is 60 notes)
    (94, 50, 96, 3, 5, True, 37),           # nothing a multiple of 4 except Hc at the fused kernel's limit; ragged row tile
    (1, 1, 1, 1, 2, False, 16),             # the smallest model the reference's flags can build
    (96, 96, 96, 16, 16, True, 64),         # every width at the fused kernel's limit
    (17, 5, 7, 2, 3, False, 33),            # everything below one 16-wide tile, odd
    (88, 88, 88, 4, 2, False, 512),         # config 2 without --use_x_prev
]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("D,H,Hc,L,Cn,use_x_prev,B", SHAPES)
def test_cl_vae_step_matches_oracle_at_other_widths(dev, D, H, Hc, L, Cn, use_x_prev, B, fused):
    from clvae_amd import _lib
    from clvae_amd.engine import VaeEngine
    assert _lib.lib().clv_vae_fused_supported(D, H, Hc, Cn, L)
    cfg = O.vae_config(original_dim=D, intermediate_dim=H, latent_dim=L, intermediate_class_dim=Hc, n_classes=Cn,
                       use_x_prev=use_x_prev, class_weight=0.8, kl_weight=0.6, w_kl_weight=0.9, w_log_var_prior=0.1)
    cfg['fused_step'] = fused
    rng = np.random.default_rng(1000 + D + H)
    p = {k: f32(v) for k, v in O.vae_init_params(cfg, seed=2).items()}
    for k in p:
        if k.endswith('bias'):
            p[k] = f32(0.05 * rng.standard_normal(p[k].shape))
    x, xp = (rng.random((B, D)) < 0.1).astype(np.float64), (rng.random((B, D)) < 0.1).astype(np.float64)
    wt = np.eye(Cn)[rng.integers(0, Cn, B)]
    ew, ez = f32(rng.standard_normal((B, Cn - 1))), f32(rng.standard_normal((B, L)))
    ref = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)
    eng = VaeEngine(cfg, B, dev)
    assert eng.fused == fused
    eng.P.set_weights(p)
    args = (T_(x, dev), T_(xp, dev), T_(wt, dev), T_(ew, dev), T_(ez, dev))
    eng.loss_and_grads(*args)
    torch.cuda.synchronize()
    got = eng.losses()
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo'):
        assert abs(got[k] - ref[k]) <= ELBO_TOL, (k, got[k], ref[k])
    assert abs(got['acc'] - ref['acc']) < 1e-6
    assert np.abs(N(eng.logits) - ref['cache']['logits']).max() < LOGIT_TOL
    g = eng.P.get_weights(eng.P.grads)
    for k in ref['grads']:
        scale = np.abs(ref['grads'][k]).max() + 1e-8
        assert np.abs(g[k] - ref['grads'][k]).max() / scale < 1e-4, k
    st = O.adam_wn_init(p)
    for _ in range(3):
        r = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)
        O.adam_wn_step(p, r['grads'], st)
        eng.loss_and_grads(*args)
        eng.P.adam_step()
    w = eng.P.get_weights()
    for k in p:
        np.testing.assert_allclose(w[k], p[k], rtol=2e-3, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("D,H,Hc", [(110, 88, 88), (97, 40, 88), (88, 97, 88), (88, 88, 120), (176, 128, 100)])
def test_cl_vae_beyond_the_fused_kernel_takes_the_chain(dev, D, H, Hc):
    """A width above 96 is outside `clv_vae_fused_supported`: the engine must pick the layer chain by itself (no flag)
    and the step must still be the oracle's."""
    from clvae_amd import _lib
    from clvae_amd.engine import VaeEngine
    B, L, Cn = 40, 4, 3
    assert not _lib.lib().clv_vae_fused_supported(D, H, Hc, Cn, L)
    cfg = O.vae_config(original_dim=D, intermediate_dim=H, latent_dim=L, intermediate_class_dim=Hc, n_classes=Cn,
                       use_x_prev=True)
    rng = np.random.default_rng(D)
    p = {k: f32(v) for k, v in O.vae_init_params(cfg, seed=3).items()}
    x, xp = (rng.random((B, D)) < 0.08).astype(np.float64), (rng.random((B, D)) < 0.08).astype(np.float64)
    wt = np.eye(Cn)[rng.integers(0, Cn, B)]
    ew, ez = f32(rng.standard_normal((B, Cn - 1))), f32(rng.standard_normal((B, L)))
    ref = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)
    eng = VaeEngine(cfg, B, dev)
    assert not eng.fused
    eng.P.set_weights(p)
    eng.loss_and_grads(T_(x, dev), T_(xp, dev), T_(wt, dev), T_(ew, dev), T_(ez, dev))
    torch.cuda.synchronize()
    got = eng.losses()
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total'):
        assert abs(got[k] - ref[k]) <= ELBO_TOL, k
    assert np.abs(N(eng.logits) - ref['cache']['logits']).max() < LOGIT_TOL
    g = eng.P.get_weights(eng.P.grads)
    for k in ref['grads']:
        assert np.abs(g[k] - ref['grads'][k]).max() / (np.abs(ref['grads'][k]).max() + 1e-8) < 1e-4, k


def _oracle_fit(p, cfg, cur, hst, wt, vcur, vhst, vwt, B, E, seed, np_seed, names):
    """The reference's fit() (cl_vae/train.py:66-71) on the oracle: np.random.shuffle permutations, Philox noise at
    (seed, step = iterations, streams 0 / 1; validation chunk j: 4 + 2j / 5 + 2j), Adam-WN per batch."""
    L, C1 = cfg['L'], cfg['C'] - 1
    keys = ('vae', 'kl_w', 'w_rec', 'kl_z')
    st = O.adam_wn_init(p)
    ref, it = {}, 0
    np.random.seed(np_seed)
    for ep in range(E):
        index = np.arange(len(cur))
        np.random.shuffle(index)
        acc = np.zeros(6)
        for b0 in range(0, len(cur), B):
            rows = index[b0:b0 + B]
            ew = f32(OP.normal(B * C1, seed, step=it, stream_id=0).reshape(B, C1))
            ez = f32(OP.normal(B * L, seed, step=it, stream_id=1).reshape(B, L))
            r = O.vae_loss_and_grads(p, cfg, cur[rows], None if hst is None else hst[rows], wt[rows], ew, ez)
            O.adam_wn_step(p, r['grads'], st)
            acc += [r['total']] + [r[k] for k in keys] + [r['acc']]
            it += 1
        logs = dict(zip(['loss'] + [n + '_loss' for n in names] + ['w_acc'], acc / (len(cur) // B)))
        acc = np.zeros(6)
        for j, b0 in enumerate(range(0, len(vcur), B)):
            ew = f32(OP.normal(B * C1, seed, step=it, stream_id=2 * (2 + j)).reshape(B, C1))
            ez = f32(OP.normal(B * L, seed, step=it, stream_id=2 * (2 + j) + 1).reshape(B, L))
            r = O.vae_loss_and_grads(p, cfg, vcur[b0:b0 + B], None if vhst is None else vhst[b0:b0 + B], vwt[b0:b0 + B],
                                     ew, ez, need_grads=False)
            acc += [r['total']] + [r[k] for k in keys] + [r['acc']]
        logs.update(dict(zip(['val_loss'] + ['val_' + n + '_loss' for n in names] + ['val_w_acc'], acc / (len(vcur) // B))))
        for k, v in logs.items():
            ref.setdefault(k, []).append(v)
    return ref


def test_train_cli_seq_length_2_tracks_an_oracle_loop(tmp_path):
    """`cl_vae/train.py r --seq_length 2 --latent_dim 3` on the real JSB_Cs (rebuilt from G7): a sample is two frames side
    by side restricted to the 59 notes that sound anywhere (original_dim 118, pinned by G9 to the reference's own lines),
    i.e. a model that is NOT 88 wide and lies beyond the fused kernel -- two epochs through `train(args)` against the
    oracle loop; `<run>.json` must say original_dim 118 so that sample.py rebuilds the same model."""
    import json
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    from clvae_amd.cl_vae import train as TR
    from clvae_amd.cl_vae.model import load_model
    from clvae_amd.initializers import init_weights
    from clvae_amd.utils.pianoroll import PianoData
    _lib.require_gpu()
    path = write_jsb_cs_pickle(str(tmp_path / 'JSB Chorales_Cs.pickle'))
    E, B, L, seed, np_seed = 2, 100, 3, 77, 5
    args = TR.build_parser().parse_args(['r', '--seq_length', '2', '--latent_dim', str(L), '--num_epochs', str(E),
                                         '--train_file', path, '--model_dir', str(tmp_path)])
    args.seed = seed
    np.random.seed(np_seed)
    model, best = TR.train(args)
    torch.cuda.synchronize()
    hist = model.history.history
    assert args.original_dim == 118 and model.engine.cfg['D'] == 118 and not model.engine.fused
    P = PianoData(path, batch_size=B, seq_length=2, step_length=1, return_y_next=False, squeeze_x=True, squeeze_y=True)
    TR.flatten_windows(P, TR.build_parser().parse_args(['r', '--seq_length', '2']))
    n_tr, n_va = len(P.x_train), len(P.x_valid)
    assert int(model.engine.P.iterations.item()) == E * (n_tr // B)
    cfg = O.vae_config(original_dim=118, latent_dim=L, n_classes=2, use_x_prev=False)
    p = {k: f32(v) for k, v in init_weights(model.engine.P.logical, model.engine.cfg, seed=seed).items()}
    ref = _oracle_fit(p, cfg, P.x_train, None, np.eye(2)[P.train_song_keys], P.x_valid, None, np.eye(2)[P.valid_song_keys],
                      B, E, seed, np_seed, ('x_decoded_mean', 'w', 'w2', 'z_args'))
    assert set(hist) == set(ref)
    worst = 0.0
    for k in sorted(ref):
        got, want = np.asarray(hist[k], np.float64), np.asarray(ref[k])
        tol = 2e-3 if k.endswith('acc') else 1e-3
        assert got.shape == (E,) and np.abs(got - want).max() <= tol, (k, got, want)
        worst = max(worst, float(np.abs(got - want).max()))
    print("cl_vae --seq_length 2 (D = 118, %d + %d samples), %d epochs: History max |gpu - oracle| %.2e"
          % (n_tr, n_va, E, worst))
    final = model.engine.P.get_weights()
    for k in p:
        d = np.abs(final[k] - p[k])
        assert float((d > 2e-3 * np.abs(p[k]) + 5e-5).mean()) <= 2e-2 and d.max() <= 2e-3 * E * (n_tr // B), k
    margs = json.load(open(os.path.join(str(tmp_path), 'r.json')))
    assert margs['original_dim'] == 118 and margs['seq_length'] == 2
    m2, _, _ = load_model(os.path.join(str(tmp_path), 'r.h5'), batch_size=1)
    assert m2.engine.cfg['D'] == 118 and m2.engine.P.get_weights()['h_w/kernel'].shape == (118, 88)


def test_train_cli_seq_length_2_with_x_prev_refuses_like_the_reference(tmp_path):
    """cl_vae/train.py:22: np.vstack of [n, 2, 88] windows and [n, 88] targets raises ValueError in the reference (G9)."""
    import clvae_amd  # noqa: F401
    from clvae_amd.cl_vae import train as TR
    path = write_jsb_cs_pickle(str(tmp_path / 'JSB Chorales_Cs.pickle'))
    args = TR.build_parser().parse_args(['r', '--seq_length', '2', '--use_x_prev', '--num_epochs', '1', '--train_file', path,
                                         '--model_dir', str(tmp_path)])
    with pytest.raises(ValueError):
        TR.train(args)


@pytest.mark.parametrize("use_x_prev", [True, False])
def test_cl_vae_enc_model_predict_matches_oracle(dev, use_x_prev):
    """enc_model = Model([x, (history)], [z_mean, w_mean]) (cl_vae/model.py:211-212,220-223): z_mean goes through the SAMPLED
    w, so the oracle gets the Philox draws predict() documents (chunk j: stream 2 * (2000 + j), step = iterations)."""
    from clvae_amd.cl_vae.model import get_model
    B, D, H, Hc, L, Cn, seed = 16, 88, 64, 40, 3, 4, 9
    model, enc = get_model(B, D, (H, L), (Hc, Cn), 'adam-wn', use_x_prev=use_x_prev, seed=seed)
    p = {k: f32(v) for k, v in model.engine.P.get_weights().items()}
    cfg = O.vae_config(original_dim=D, intermediate_dim=H, latent_dim=L, intermediate_class_dim=Hc, n_classes=Cn,
                       use_x_prev=use_x_prev)
    rng = np.random.default_rng(4)
    x, xp = (rng.random((3 * B, D)) < 0.06).astype(np.float64), (rng.random((3 * B, D)) < 0.06).astype(np.float64)
    z_mean, w_mean = enc.predict([x, xp] if use_x_prev else x)
    assert z_mean.shape == (3 * B, L) and w_mean.shape == (3 * B, Cn - 1)
    for j in range(3):
        ew = f32(OP.normal(B * (Cn - 1), seed, step=0, stream_id=2 * (2000 + j)).reshape(B, Cn - 1))
        c = O.vae_forward(p, cfg, x[j * B:(j + 1) * B], xp[j * B:(j + 1) * B], ew, np.zeros((B, L)))
        np.testing.assert_allclose(w_mean[j * B:(j + 1) * B], c['w_mean'], atol=2e-5)
        np.testing.assert_allclose(z_mean[j * B:(j + 1) * B], c['z_mean'], atol=2e-5)
    with pytest.raises(ValueError):
        enc.predict([x[:B + 1], xp[:B + 1]] if use_x_prev else x[:B + 1])


@pytest.mark.parametrize("use_x_prev", [True, False])
def test_cl_vrnn_encoder_predict_matches_oracle(dev, use_x_prev):
    """encoder = Model(X, [Z_mean, Z_log_var, W]) (cl_vrnn/model.py:266): W is the sampled label (Philox chunk j: stream
    2 * (1000 + j)), the Z heads follow the encoder LSTM on [X, W]."""
    from clvae_amd.cl_vrnn.model import get_model
    B, Tn, L, Cn, seed = 4, 12, 2, 10, 13
    model, encoder = get_model(B, 88, 88, L, Tn, Cn, use_x_prev, 'adam-wn', seed=seed)
    p = {k: f32(v) for k, v in model.engine.P.get_weights().items()}
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=use_x_prev)
    rng = np.random.default_rng(6)
    X = (rng.random((2 * B, Tn, 88)) < 0.05).astype(np.float64)
    Zm, Zlv, W = encoder.predict(X)
    assert Zm.shape == (2 * B, Tn, L) and Zlv.shape == (2 * B, Tn, L) and W.shape == (2 * B, Cn)
    for j in range(2):
        eW = f32(OP.normal(B * (Cn - 1), seed, step=0, stream_id=2 * (1000 + j)).reshape(B, Cn - 1))
        rows = slice(j * B, (j + 1) * B)
        c = O.vrnn_forward(p, cfg, X[rows], np.zeros_like(X[rows]), eW, np.zeros((B, Tn, L)))
        np.testing.assert_allclose(W[rows], c['W'], atol=2e-6)
        np.testing.assert_allclose(Zm[rows], c['Z_mean'], atol=3e-5)
        np.testing.assert_allclose(Zlv[rows], c['Z_log_var'], atol=3e-5)
