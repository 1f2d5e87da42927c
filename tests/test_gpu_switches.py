"""-m gpu: every kernel-selection switch of VrnnEngine (engine.SWITCHES) in its NON-default position, through a two-step
oracle check -- loss terms, gradients of the first step, parameters after two Adam-WN steps.  The defaults are what every other
test and the bench run; this file is what keeps the other positions (fallback chains, measurement knobs) honest."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import clvae_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def _switch_names():
    import clvae_amd  # noqa: F401
    from clvae_amd.engine import SWITCHES
    return sorted(SWITCHES)


# the batch a switch needs to matter: the large-batch kernels start at 768 rows, the dense hW forward at 512
LARGE = {'lstm_mx': (768, 3, 12), 'frames_u8': (768, 3, 12), 'dense_hw_fwd': (768, 3, 2)}
# what else must be off for a switch to select anything (the latent head only exists outside the pair kernels)
ALSO = {'fuse_latent': {'fuse_pair': False}}


@pytest.mark.parametrize("key", _switch_names())
def test_switch_in_its_other_position_still_follows_the_oracle(dev, key):
    from clvae_amd.engine import SWITCHES, VrnnEngine
    B, Tn, L = LARGE.get(key, (6, 5, 2))
    Cn = 4
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True, class_weight=0.8, kl_weight=0.7,
                        w_kl_weight=0.9, w_log_var_prior=0.1)
    rng = np.random.default_rng(len(key))
    p = {k: f32(v) for k, v in O.vrnn_init_params(cfg, seed=2).items()}
    win = (rng.random((B, Tn + 1, 88)) < 0.05).astype(np.float64)
    X, Xp = win[:, 1:].copy(), win[:, :-1].copy()
    wt = np.eye(Cn)[rng.integers(0, Cn, B)]
    eW, eZ = f32(rng.standard_normal((B, Cn - 1))), f32(rng.standard_normal((B, Tn, L)))
    default = SWITCHES[key][1]
    eng = VrnnEngine(dict(cfg, **ALSO.get(key, {}), **{key: not default}), B, dev)
    base = VrnnEngine(dict(cfg, **ALSO.get(key, {})), B, dev)
    attr = {'lstm_mx': 'use_mx'}.get(key, key)
    # the switch is live at this shape: the default engine has it in the default position, this one has not
    assert bool(getattr(base, attr)) == default and bool(getattr(eng, attr)) == (not default), (key, getattr(base, attr), getattr(eng, attr))
    del base
    eng.frames_exact_bf16 = True          # (0 / 1 frames: what TrainStep notes when it stages bytes; lets the dense hW products run)
    eng.P.set_weights(p)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
    args = (t(X), t(Xp), t(wt), t(eW), t(eZ.reshape(B * Tn, L)))
    st = O.adam_wn_init(p)
    for step in range(2):
        ref = O.vrnn_loss_and_grads(p, cfg, X, Xp, wt, eW, eZ)
        eng.loss_and_grads(*args)
        torch.cuda.synchronize()
        got = eng.losses()
        for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total'):
            assert abs(got[k] - ref[k]) <= 1e-3, (key, step, k, got[k], ref[k])
        if step == 0:
            g = eng.P.get_weights(eng.P.grads)
            for k in ref['grads']:
                scale = np.abs(ref['grads'][k]).max() + 1e-8
                assert np.abs(g[k] - ref['grads'][k]).max() / scale < 3e-4, (key, k)
        O.adam_wn_step(p, ref['grads'], st)
        eng.P.adam_step()
    w = eng.P.get_weights()
    for k in p:      # (an entry whose gradient is within rounding of zero moves by +-lr per step whatever its size: tests/test_gpu_timed_step.py)
        d = np.abs(w[k] - p[k])
        assert float((d > 2e-3 * np.abs(p[k]) + 5e-5).mean()) <= 2e-2 and d.max() <= 4.1e-3, (key, k, d.max())
