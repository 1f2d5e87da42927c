"""G6: numbers produced by the REFERENCE'S OWN LINES (tests/golden/make_g6_reference_lines.py exec'd
utils/weightnorm.py:75-178 and the loss closures cl_vae/model.py:193-196, 202-206, cl_vrnn/model.py:236-239, 247-252 under
a numpy namespace, in the build container).  Both the oracle (here, CPU) and the HIP kernels (the -m gpu tests below) are
held to them: for these two slices of the arithmetic the parity is pinned by a reference run, not by a restatement.
Layer semantics (Dense / LSTM / TimeDistributed / keras.losses) stay [K]-recalled: DESIGN.md section 2."""
import numpy as np
import pytest

from helpers import golden
from oracle import clvae_oracle as O

G = golden("g6_reference_lines.npz")
NAMES = [str(n) for n in G['opt/names']]
NSTEPS = 4


def _opt_inputs():
    p = {n: G['opt/p0/' + n].astype(np.float64) for n in NAMES}
    grads = [{n: G['opt/g%d/%s' % (s, n)].astype(np.float64) for n in NAMES} for s in range(NSTEPS)]
    return p, grads


def test_oracle_adam_wn_matches_the_reference_lines():
    """oracle.adam_wn_step == AdamWithWeightnorm.get_updates (utils/weightnorm.py:75-143) + its helpers (:146-178), four
    steps, gradients spanning six orders of magnitude, zero gradient rows; parameters after every step and the whole
    optimizer state at the end."""
    p, grads = _opt_inputs()
    st = O.adam_wn_init(p)
    for s in range(NSTEPS):
        O.adam_wn_step(p, grads[s], st)
        for n in NAMES:
            np.testing.assert_allclose(p[n], G['opt/p%d/%s' % (s + 1, n)], rtol=1e-12, atol=1e-15, err_msg="%s step %d" % (n, s))
    # state in the reference's creation order: ms, vs (one per parameter), then (V_scaler, m_g, v_g) per matrix
    k = len(NAMES)
    for i, n in enumerate(NAMES):
        np.testing.assert_allclose(st['m'][n], G['opt/state/%02d' % i], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(st['v'][n], G['opt/state/%02d' % (k + i)], rtol=1e-12, atol=1e-300)
    j = 2 * k
    for n in NAMES:
        if p[n].ndim > 1:
            np.testing.assert_allclose(st['s'][n], G['opt/state/%02d' % j], rtol=1e-12)
            np.testing.assert_allclose(st['mg'][n], G['opt/state/%02d' % (j + 1)], rtol=1e-12, atol=1e-300)
            np.testing.assert_allclose(st['vg'][n], G['opt/state/%02d' % (j + 2)], rtol=1e-12, atol=1e-300)
            j += 3
    assert j == int(G['opt/n_state']) and st['t'] == NSTEPS


def test_oracle_kl_terms_match_the_reference_closures():
    """kl_loss / w_kl_loss of both models as the reference's closures computed them (three priors)."""
    za = G['loss/vae/z_args']
    L = za.shape[1] // 2
    np.testing.assert_allclose(O.kl_gauss(za[:, :L], za[:, L:])[0], G['loss/vae/kl_z'], rtol=1e-13)
    Za = G['loss/vrnn/Z_args']
    L = Za.shape[2] // 2
    np.testing.assert_allclose(O.kl_gauss(Za[..., :L], Za[..., L:])[0], G['loss/vrnn/kl_z'], rtol=1e-13)
    for prior in (0.0, 0.5, -1.0):
        np.testing.assert_allclose(O.kl_w_prior(G['loss/vae/w_mean'], G['loss/vae/w_log_var'], prior)[0],
                                   G['loss/vae/kl_w/prior%g' % prior], rtol=1e-13)
        np.testing.assert_allclose(O.kl_w_prior(G['loss/vrnn/W_mean'], G['loss/vrnn/W_log_var'], prior)[0],
                                   G['loss/vrnn/kl_w/prior%g' % prior], rtol=1e-13)


# ------------------------------------------------------------------------------------------------- HIP kernels vs G6
@pytest.fixture(scope="module")
def dev():
    torch = pytest.importorskip("torch")
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def _t(a, dev):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


@pytest.mark.gpu
@pytest.mark.parametrize("fast", [False, True])
def test_hip_adam_wn_matches_the_reference_lines(dev, fast):
    """clv_adam_wn_step (five launches) and clv_adam_wn_step with known column sums (two launches: the form the timed
    step uses) against the parameters the reference's optimizer lines produced.  fp32 on the device: rtol 2e-5."""
    import torch
    from clvae_amd.engine import FlatParams
    p, grads = _opt_inputs()
    shapes = [(n, p[n].shape) for n in NAMES]
    P = FlatParams(shapes, dev, pre=5)
    P.set_weights(p)
    for s in range(NSTEPS):
        for n in NAMES:
            P.g(n).copy_(_t(grads[s][n], dev))
        gdot = None
        if fast and s > 0 and P.norms_valid:       # the first tensor's sum_j W dW per column, as the backward pass leaves it
            gdot = _t((P.get_weights()[NAMES[0]] * grads[s][NAMES[0]]).sum(0), dev)
        P.adam_step(gdot=gdot)
        torch.cuda.synchronize()
        got = P.get_weights()
        for n in NAMES:
            np.testing.assert_allclose(got[n], G['opt/p%d/%s' % (s + 1, n)], rtol=2e-5, atol=2e-7, err_msg="%s step %d" % (n, s))
    assert int(P.iterations.item()) == NSTEPS


@pytest.mark.gpu
def test_hip_kl_kernels_match_the_reference_closures(dev):
    """clv_gauss_fwd's row KL (both models' kl_loss) and clv_label_fwd's KL term (w_kl_loss with the prior) against the
    closures' outputs."""
    import torch
    from clvae_amd import ops
    for key, zkey in (('vae', 'z_args'), ('vrnn', 'Z_args')):
        za = G['loss/%s/%s' % (key, zkey)]
        L = za.shape[-1] // 2
        rows = za.reshape(-1, 2 * L)
        R = rows.shape[0]
        z = torch.zeros(R, L, device=dev); kl = torch.empty(R, device=dev)
        ops.gauss_fwd(R, L, _t(rows, dev), _t(np.zeros((R, L)), dev), z, L, kl)
        torch.cuda.synchronize()
        np.testing.assert_allclose(kl.cpu().numpy().astype(np.float64), G['loss/%s/kl_z' % key].reshape(-1), rtol=2e-5, atol=2e-6)
    for key, mk, lk in (('vae', 'w_mean', 'w_log_var'), ('vrnn', 'W_mean', 'W_log_var')):
        m, lv = G['loss/%s/%s' % (key, mk)], G['loss/%s/%s' % (key, lk)]
        B, C1 = m.shape
        wargs = _t(np.concatenate([m, lv], 1), dev)
        y = _t(np.eye(C1 + 1)[np.arange(B) % (C1 + 1)], dev)
        for prior in (0.0, 0.5, -1.0):
            w = torch.empty(B, C1 + 1, device=dev); rl = torch.empty(B, 3, device=dev)
            ops.label_fwd(B, C1 + 1, wargs, wargs[:, C1:], 2 * C1, _t(np.zeros((B, C1)), dev), y, prior, w, rl)
            torch.cuda.synchronize()
            np.testing.assert_allclose(rl.cpu().numpy()[:, 0].astype(np.float64), G['loss/%s/kl_w/prior%g' % (key, prior)],
                                       rtol=2e-5, atol=2e-6)
