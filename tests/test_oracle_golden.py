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
