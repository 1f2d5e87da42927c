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


# TrainStep's own run-time switches (environment variables read when the step object is built): each in its non-default
# position, three captured steps fed by the bound-batch cursor against an oracle loop with the Philox noise of the same keys.
TRAIN_SWITCHES = [
    {'CLV_STAGE_IN_LABEL': '0'},                                   # the gather launch instead of the assembly inside the label launch
    {'CLV_FRONT_FUSED': '0'},                                      # the frame projections as a launch of their own behind the label launch
    {'CLV_FAST_ADAM': '0'},                                        # the five-launch Adam-WN chain instead of the two-launch form
    {'CLV_FORCE_DP_GRAPHS': '1'},                                  # the data-parallel schedule (no-op collectives) on one GPU
    {'CLV_FORCE_DP_GRAPHS': '1', 'CLV_DP_EAGER_UPDATE': '0'},      # ... with the optimizer pieces as graphs of their own
    {'CLV_FORCE_DP_GRAPHS': '1', 'CLV_FINE_GRID': '1'},            # ... with the finer weight-gradient grid
    {'CLV_CAPTURE_COLLECTIVES': '0', 'CLV_FORCE_DP_GRAPHS': '1'},  # the split schedule asked for by name
]


@pytest.mark.parametrize("env", TRAIN_SWITCHES, ids=lambda e: "+".join("%s=%s" % kv for kv in sorted(e.items())))
def test_trainstep_switch_in_its_other_position_still_follows_the_oracle(dev, monkeypatch, env):
    from oracle import philox as OP
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    B, Tn, L, Cn, steps, seed = 16, 10, 2, 10, 3, 808
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(3)
    p = {k: f32(v) for k, v in O.vrnn_init_params(cfg, seed=9).items()}
    win = (rng.random((steps * B, Tn + 1, 88)) < 0.05)
    keys = np.eye(Cn)[rng.integers(0, Cn, steps * B)]
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=seed)
    u8 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)
    ts.bind_batches(u8(win[:, 1:].reshape(steps * B, -1)), u8(win[:, :-1].reshape(steps * B, -1)),
                    torch.as_tensor(keys.astype(np.float32), device=dev), idx=None, period=steps, stride=B)
    st = O.adam_wn_init(p)
    for it in range(steps):
        ts.step()
        torch.cuda.synchronize()
        got = eng.losses()
        sl = slice(it * B, (it + 1) * B)
        eW = f32(OP.normal(B * (Cn - 1), seed, step=it, stream_id=0).reshape(B, Cn - 1))
        eZ = f32(OP.normal(B * Tn * L, seed, step=it, stream_id=1).reshape(B, Tn, L))
        ref = O.vrnn_loss_and_grads(p, cfg, win[sl, 1:].astype(np.float64), win[sl, :-1].astype(np.float64), keys[sl], eW, eZ)
        O.adam_wn_step(p, ref['grads'], st)
        for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total'):
            assert abs(got[k] - ref[k]) <= 1e-3, (env, it, k, got[k], ref[k])
    assert int(eng.P.iterations.item()) == steps
    w = eng.P.get_weights()
    for k in p:
        d = np.abs(w[k] - p[k])
        assert float((d > 1e-4 * np.abs(p[k]) + 2e-5).mean()) <= 1e-2 and d.max() <= 2e-3 * steps, (env, k, d.max())
