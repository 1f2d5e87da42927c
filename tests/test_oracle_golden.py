"""G4: the oracle reproduces its committed outputs (guards the checker itself against drift)."""
import numpy as np

from helpers import golden
from oracle import clvae_oracle as O

G = golden("g4_oracle_steps.npz")


def _params(prefix):
    return {k[len(prefix):]: G[k].astype(np.float64) for k in G.files if k.startswith(prefix)}


def test_vae_step_reproduces_golden():
    cfg = O.vae_config(latent_dim=4, n_classes=2, use_x_prev=True)
    p = _params('vae/p/')
    r = O.vae_loss_and_grads(p, cfg, G['vae/x'].astype(float), G['vae/xp'].astype(float), G['vae/wt'], G['vae/ew'], G['vae/ez'])
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo', 'acc'):
        assert abs(float(G['vae/loss/' + k]) - r[k]) < 1e-12, k
    for k, v in r['grads'].items():
        np.testing.assert_allclose(G['vae/g/' + k], v, rtol=1e-12, atol=1e-15)
    st = O.adam_wn_init(p)
    for _ in range(3):
        rr = O.vae_loss_and_grads(p, cfg, G['vae/x'].astype(float), G['vae/xp'].astype(float), G['vae/wt'], G['vae/ew'], G['vae/ez'])
        O.adam_wn_step(p, rr['grads'], st)
    for k in p:
        np.testing.assert_allclose(G['vae/p3/' + k], p[k], rtol=1e-10, atol=1e-13)


def test_vrnn_step_reproduces_golden():
    cfg = O.vrnn_config(latent_dim=2, seq_length=16, n_classes=10, use_x_prev=True)
    p = _params('vrnn/p/')
    r = O.vrnn_loss_and_grads(p, cfg, G['vrnn/X'].astype(float), G['vrnn/Xp'].astype(float), G['vrnn/wt'], G['vrnn/eW'], G['vrnn/eZ'])
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo'):
        assert abs(float(G['vrnn/loss/' + k]) - r[k]) < 1e-12, k
    np.testing.assert_allclose(G['vrnn/c/logits'], r['cache']['logits'], atol=1e-12)
    for k, v in r['grads'].items():
        np.testing.assert_allclose(G['vrnn/g/' + k], v, rtol=2e-6, atol=1e-9)      # big tensors are stored as f32


# --------------------------------------------------------------------------- #
# G4-independent: the oracle against a SECOND derivation of the same graphs (tests/golden/make_g4_independent.py: float64
# torch.autograd written from the reference's model.py lines, no import of oracle/).  This is the pin that is not the
# oracle grading itself: losses, per-note logits and every gradient tensor of both models, on real JSB frames, with weights
# that drive logits through both Bernoulli clip points and gates through the flat regions of the hard sigmoid.
# --------------------------------------------------------------------------- #
GI = golden("g4_independent.npz")


def _gi(prefix):
    return {k[len(prefix):]: GI[k].astype(np.float64) for k in GI.files if k.startswith(prefix)}


def _check_grads(got, prefix, tol):
    want = _gi(prefix)
    assert set(got) == set(want)
    for k, v in got.items():
        scale = max(np.abs(want[k]).max(), 1e-30)
        assert np.abs(v - want[k]).max() <= tol * scale, (k, np.abs(v - want[k]).max() / scale)


def test_vae_oracle_matches_the_independent_autograd_derivation():
    cw, kw, wkw = GI['vae/wts']
    cfg = O.vae_config(latent_dim=4, n_classes=2, use_x_prev=True, class_weight=cw, kl_weight=kw, w_kl_weight=wkw,
                       w_log_var_prior=float(GI['vae/prior']))
    r = O.vae_loss_and_grads(_gi('vae/p/'), cfg, GI['vae/x'].astype(float), GI['vae/xp'].astype(float), GI['vae/wt'],
                             GI['vae/ew'].astype(float), GI['vae/ez'].astype(float))
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'acc'):
        assert abs(float(GI['vae/loss/' + k]) - r[k]) < 1e-10, (k, float(GI['vae/loss/' + k]), r[k])
    np.testing.assert_allclose(r['cache']['logits'], GI['vae/logits'], atol=1e-11)
    assert (np.abs(GI['vae/logits']) > 16.2).any()                 # the fixture does reach beyond both clip points
    _check_grads(r['grads'], 'vae/g/', 1e-9)


def test_vrnn_oracle_matches_the_independent_autograd_derivation():
    cw, kw, wkw = GI['vrnn/wts']
    T = GI['vrnn/X'].shape[1]
    cfg = O.vrnn_config(latent_dim=2, seq_length=T, n_classes=10, use_x_prev=True, class_weight=cw, kl_weight=kw,
                        w_kl_weight=wkw, w_log_var_prior=float(GI['vrnn/prior']))
    r = O.vrnn_loss_and_grads(_gi('vrnn/p/'), cfg, GI['vrnn/X'].astype(float), GI['vrnn/Xp'].astype(float), GI['vrnn/wt'],
                              GI['vrnn/eW'].astype(float), GI['vrnn/eZ'].astype(float))
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'acc'):
        assert abs(float(GI['vrnn/loss/' + k]) - r[k]) < 1e-10, (k, float(GI['vrnn/loss/' + k]), r[k])
    np.testing.assert_allclose(r['cache']['logits'], GI['vrnn/logits'], atol=1e-10)
    np.testing.assert_allclose(r['cache']['enc_h'], GI['vrnn/enc_h'], atol=1e-6)      # stored as f32
    np.testing.assert_allclose(r['cache']['dec_h'], GI['vrnn/dec_h'], atol=1e-6)
    assert (GI['vrnn/logits'] > 16.0).any() and (GI['vrnn/logits'] < -16.2).any()
    _check_grads(r['grads'], 'vrnn/g/', 2e-7)                      # the large tensors are stored as f32
