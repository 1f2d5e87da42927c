"""G2/G3: the product's numpy samplers and generate_sample loops vs outputs of the reference's own
functions (exec'd from cl_vae/model.py:9-74 and cl_vrnn/model.py:9-96 under fixed seeds)."""
import numpy as np
import pytest

import clvae_amd  # noqa: F401
from clvae_amd.cl_vae import model as vae
from clvae_amd.cl_vrnn import model as vrnn
from helpers import StubModel, golden

G = golden("g2_g3_samplers.npz")
mu, lv, p = G['g2/mu'], G['g2/lv'], G['g2/p']


@pytest.mark.parametrize("name,ns", [('vae', vae), ('vrnn', vrnn)])
@pytest.mark.parametrize("seed", [0, 7])
def test_samplers(name, ns, seed):
    def chk(key, fn):
        np.random.seed(seed)
        np.testing.assert_array_equal(G['g2/%s/%s/%d' % (name, key, seed)], fn())
    chk('sample_x', lambda: ns.sample_x(p))
    chk('sample_w', lambda: ns.sample_w((mu, lv)))
    chk('sample_w_nonoise', lambda: ns.sample_w((mu, lv), add_noise=False))
    chk('sample_w_n3', lambda: ns.sample_w((mu, lv), nsamps=3))
    chk('sample_w_nrm', lambda: ns.sample_w((mu, lv), nrm_samp=True))
    chk('sample_z', lambda: ns.sample_z((mu[:, :4], lv[:, :4])))
    chk('sample_z_n2', lambda: ns.sample_z((mu[:, :4], lv[:, :4]), nsamps=2))


def test_sample_w_discrete():
    np.random.seed(3)
    np.testing.assert_array_equal(G['g2/vrnn/sample_w_discrete/3'], vrnn.sample_w_discrete(np.array([0.1, 0.2, 0.3, 0.4])))


@pytest.mark.parametrize("tag,kw", [('infer_w', dict(w_val=None, use_x_prev=True)),
                                    ('given_w', dict(w_val=np.array([[0.25, 0.75]]), use_x_prev=False)),
                                    ('z_prior', dict(w_val=None, use_z_prior=True, use_x_prev=True, w_sample=True))])
def test_generate_sample_cl_vae(tag, kw):
    log = []
    dec, wenc, zenc = StubModel('dec', (88, 2), log), StubModel('w_enc', (1,), log), StubModel('z_enc', (4, 2), log)
    x_seed = (np.arange(88) % 11 == 0).astype(float)
    np.random.seed(11)
    Xs = vae.generate_sample(dec, wenc, zenc, x_seed, 6, **kw)
    np.testing.assert_array_equal(G['g3/vae/%s/Xs' % tag], Xs)
    ncalls = [sum(1 for l in log if l[0] == k) for k in ('dec', 'w_enc', 'z_enc')]
    np.testing.assert_array_equal(G['g3/vae/%s/ncalls' % tag], ncalls)


@pytest.mark.parametrize("tag,kw", [('infer_w', dict(w_val=None, seq_length=4)),
                                    ('discrete_w', dict(w_val=None, seq_length=4, w_discrete=True)),
                                    ('given_w', dict(w_val=np.eye(10)[3][None, :], seq_length=4)),
                                    ('no_x_prev', dict(w_val=np.eye(10)[1][None, :], seq_length=4))])
def test_generate_sample_cl_vrnn(tag, kw):
    log = []
    dec, wenc, zenc = StubModel('dec', (88, 3), log), StubModel('w_enc', (9,), log), StubModel('z_enc', (2, 3), log)
    x_seed = G['g3/vrnn/x_seed']
    np.random.seed(13)
    Xs = vrnn.generate_sample(dec, wenc, zenc, x_seed, 5, tag != 'no_x_prev', **kw)
    np.testing.assert_array_equal(G['g3/vrnn/%s/Xs' % tag], Xs)
    ncalls = [sum(1 for l in log if l[0] == k and l[1] != 'reset') for k in ('dec', 'w_enc', 'z_enc')]
    np.testing.assert_array_equal(G['g3/vrnn/%s/ncalls' % tag], ncalls)
    np.testing.assert_array_equal(G['g3/vrnn/%s/nreset' % tag], [dec.n_reset, wenc.n_reset, zenc.n_reset])
