"""-m gpu: the step bench.py TIMES, as it is timed, against the oracle at the sizes it is timed at.

tests/test_gpu_models.py checks `loss_and_grads` with injected noise and steps the optimizer separately.  What the
benchmark (and `fit()`) replays is something else: a captured hipGraph whose kernels draw their own Philox noise, pack the
recurrent weights inside the label launch, run the label path's backward inside the pair backward kernel, stage uint8
frames, and update with the two-launch Adam-with-weight-norm from sums the backward pass left behind.  Here that exact
path -- TrainStep(use_graph=True) with every default -- runs for a few steps on fresh batches, and an oracle loop
(`vrnn_loss_and_grads` / `vae_loss_and_grads` + `adam_wn_step`, cl_vrnn/train.py:66-71, utils/weightnorm.py:75-143) fed the
noise of oracle/philox.py at the same (seed, step, stream) follows it: every loss term of every step to 1e-3, the
parameters after the last step.

Tolerance of the parameters.  One Adam step moves an entry by lr * m_hat / (sqrt(v_hat) + eps) = +-lr for ANY gradient
that is not ~0, so two correct implementations can differ by up to 2 lr = 2e-3 (absolute) per step on an entry whose
gradient is within rounding of zero (its sign decides the direction), and agree to rounding everywhere else.  The bar is
therefore: at most a 1e-4 fraction of a tensor's entries may differ by more than rtol 1e-4 / atol 2e-5 (measured on
MI355X: max |dw| 8e-6 after three steps at 256 x 128, 5e-6 after two at 1024 x 256), and none by more than 2 lr per step
taken.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import clvae_oracle as O
from oracle import philox as OP

pytestmark = pytest.mark.gpu

LOSS_TOL = 1e-3
LR = 1e-3


@pytest.fixture(scope="module")
def dev():
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def u8(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)


def ft(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def check_params(got, want, steps, rtol=1e-4, atol=2e-5, frac=1e-4, what=""):
    worst = 0.0
    for k in want:
        d = np.abs(got[k] - want[k])
        off = d > rtol * np.abs(want[k]) + atol
        assert off.mean() <= frac, "%s %s: %.2e of the entries beyond rtol %g / atol %g" % (what, k, off.mean(), rtol, atol)
        assert d.max() <= 2 * LR * steps, "%s %s: max |dw| %.2e" % (what, k, d.max())
        worst = max(worst, float(d.max()))
    return worst


@pytest.mark.parametrize("feed", ["stage_batch", "cursor"])
@pytest.mark.parametrize("B,Tn,L,steps", [(256, 128, 2, 3),        # BASELINE config 3 (config 4 per GPU): bench.py's default
                                          (1024, 256, 32, 2)])     # config 5 per GPU
def test_cl_vrnn_timed_step_tracks_the_oracle(dev, B, Tn, L, steps, feed):
    """feed = "cursor": what bench.py and Model.fit do -- TrainStep.bind_batches, the mini-batch assembled INSIDE the
    captured step from the device-resident data set by the device step counter (at 256 x 128: inside the label forward
    launch), nothing staged from the host; "stage_batch": one staging launch per step in front of the graph."""
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep
    Cn, seed = 10, 4321
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(B + L)
    p = {k: f32(v) for k, v in O.vrnn_init_params(cfg, seed=5).items()}
    win = (rng.random((steps * B, Tn + 1, 88)) < 0.0443)
    keys = np.eye(Cn)[rng.integers(0, Cn, steps * B)]
    eng = VrnnEngine(cfg, B, dev)
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=seed)             # every default: graph, in-kernel noise, fast Adam, label-in-pair
    assert ts.use_graph and ts.fast_adam and ts.ar is None
    Xd, Xpd, wd = u8(win[:, 1:], dev), u8(win[:, :-1], dev), ft(keys, dev)
    st = O.adam_wn_init(p)
    if feed == "cursor":
        ts.bind_batches(Xd.reshape(steps * B, -1), Xpd.reshape(steps * B, -1), wd, idx=None, period=steps, stride=B)
        assert (ts._label_stage() is not None) == (L <= 8)        # the pair path's label launch assembles the batch itself
    for it in range(steps):
        sl = slice(it * B, (it + 1) * B)
        if feed == "stage_batch":
            ts.stage_batch(Xd[sl], Xpd[sl], wd[sl])
        ts.step()
        torch.cuda.synchronize()
        got = eng.losses()
        eW = f32(OP.normal(B * (Cn - 1), seed, step=it, stream_id=0).reshape(B, Cn - 1))
        eZ = f32(OP.normal(B * Tn * L, seed, step=it, stream_id=1).reshape(B, Tn, L))
        ref = O.vrnn_loss_and_grads(p, cfg, win[sl, 1:].astype(np.float64), win[sl, :-1].astype(np.float64), keys[sl], eW, eZ)
        O.adam_wn_step(p, ref['grads'], st)
        print("cl_vrnn %dx%d L=%d %s step %d (%s): total gpu %.6f oracle %.6f |d| %.2e" %
              (B, Tn, L, feed, it, "eager" if it == 0 else "graph replay", got['total'], ref['total'], abs(got['total'] - ref['total'])))
        for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo'):
            assert abs(got[k] - ref[k]) <= LOSS_TOL, (it, k, got[k], ref[k])
    # the path that ran is the one bench.py times
    assert ts._graphs is not None and len(ts._graphs) == 1 and eng.frames_exact_bf16
    assert eng.P.norms_valid and int(eng.P.iterations.item()) == steps
    if L <= 8:
        assert eng.fuse_pair and eng.label_in_pair and eng.folds_noise()
    else:
        assert eng.use_mx and eng.fuse_latent
    worst = check_params(eng.P.get_weights(), p, steps, what="cl_vrnn %dx%d" % (B, Tn))
    print("cl_vrnn %dx%d: parameters after %d steps, max |dw| %.2e" % (B, Tn, steps, worst))


@pytest.mark.parametrize("bf16", [False, True])
def test_cl_vae_timed_step_tracks_the_oracle(dev, bf16):
    """BASELINE config 2: the fused cl_vae step (one launch: 6 Dense layers forward and backward, Philox noise inside,
    loss means and the step counter in the slab-sum launch) + Adam-WN, replayed as a graph, batch 512.  bf16=True rounds
    the operands of every product to bf16 (tests/test_gpu_models.py::test_cl_vae_bf16_step_tolerance states its bars
    for one step; three steps of it stay inside 2e-3 on every loss term)."""
    from clvae_amd.engine import VaeEngine
    from clvae_amd.trainer import TrainStep
    B, L, Cn, steps, seed = 512, 4, 2, 3, 99
    cfg = O.vae_config(latent_dim=L, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(7)
    p = {k: f32(v) for k, v in O.vae_init_params(cfg, seed=2).items()}
    fr = (rng.random((steps * B, 2, 88)) < 0.0443)
    keys = np.eye(Cn)[rng.integers(0, Cn, steps * B)]
    eng = VaeEngine(dict(cfg, bf16=bf16), B, dev)
    assert eng.fused
    eng.P.set_weights(p)
    ts = TrainStep(eng, seed=seed)
    xd, xpd, wd = u8(fr[:, 1], dev), u8(fr[:, 0], dev), ft(keys, dev)
    st = O.adam_wn_init(p)
    tol = 2e-3 if bf16 else LOSS_TOL
    for it in range(steps):
        sl = slice(it * B, (it + 1) * B)
        ts.stage_batch(xd[sl], xpd[sl], wd[sl])
        ts.step()
        torch.cuda.synchronize()
        got = eng.losses()
        ew = f32(OP.normal(B * (Cn - 1), seed, step=it, stream_id=0).reshape(B, Cn - 1))
        ez = f32(OP.normal(B * L, seed, step=it, stream_id=1).reshape(B, L))
        ref = O.vae_loss_and_grads(p, cfg, fr[sl, 1].astype(np.float64), fr[sl, 0].astype(np.float64), keys[sl], ew, ez)
        O.adam_wn_step(p, ref['grads'], st)
        print("cl_vae 512 %s step %d: total gpu %.6f oracle %.6f |d| %.2e" %
              ("bf16" if bf16 else "fp32", it, got['total'], ref['total'], abs(got['total'] - ref['total'])))
        for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo'):
            assert abs(got[k] - ref[k]) <= tol, (it, k, got[k], ref[k])
    assert ts._graphs is not None and ts._folded() and int(eng.P.iterations.item()) == steps
    if bf16:       # products rounded to 8 bits: the parameters follow to a few percent of a step
        check_params(eng.P.get_weights(), p, steps, rtol=5e-2, atol=5e-4, frac=2e-2, what="cl_vae bf16")
    else:
        check_params(eng.P.get_weights(), p, steps, what="cl_vae fp32")


@pytest.mark.parametrize("source", ["rows", "windows", "windows-no-history"])
def test_batch_assembly_inside_the_label_launch_equals_the_gather_launch(dev, monkeypatch, source):
    """TrainStep.bind_batches: the label forward launch assembles the mini-batch itself (clv_vrnn_label_fwd_x(stage)) where it
    can -- byte frames, the fused pair path.  Against the same steps with the gather launch (CLV_STAGE_IN_LABEL=0): the staged
    X / history frames / labels, every loss and every parameter after three replayed steps, bit for bit.  `rows`: whole
    rows in two byte tensors (bench.py); `windows`: overlapping windows of one frame store through a start table
    (Model.fit: utils.pianoroll.Windows), shuffled by a row list."""
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep, DevWindows
    B, Tn, L, Cn, nb = 16, 12, 2, 10, 3
    hist_on = source != "windows-no-history"         # a decoder without history frames (--no use_x_prev)
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=hist_on)
    rng = np.random.default_rng(31)
    p = {k: f32(v) for k, v in O.vrnn_init_params(cfg, seed=6).items()}
    n = nb * B
    keys = ft(np.eye(Cn)[rng.integers(0, Cn, n)], dev)
    if source == "rows":
        win = (rng.random((n, Tn + 1, 88)) < 0.05)
        cur, hist, idx = u8(win[:, 1:].reshape(n, -1), dev), u8(win[:, :-1].reshape(n, -1), dev), None
    else:
        store = u8(rng.random((n + Tn + 40, 88)) < 0.05, dev)
        starts = torch.as_tensor(rng.permutation(n + 30)[:n].astype(np.int64), device=dev)
        cur, hist = DevWindows(store, starts, 1), (DevWindows(store, starts, 0) if hist_on else None)
        idx = torch.as_tensor(rng.permutation(n).astype(np.int64), device=dev)
    runs = {}
    for staged in ("1", "0"):
        monkeypatch.setenv("CLV_STAGE_IN_LABEL", staged)
        eng = VrnnEngine(cfg, B, dev)
        eng.P.set_weights(p)
        ts = TrainStep(eng, seed=77)
        ts.bind_batches(cur, hist, keys, idx=idx, period=nb, stride=B)
        assert (ts._label_stage() is not None) == (staged == "1")
        out = []
        for it in range(3):
            ts.step()
            torch.cuda.synchronize()
            # what the step was fed: the float batch, or (round 6, the label launch's own stage) the byte batch the later launches read
            f8 = ts._f8
            assert (f8 is not None) == (staged == "1")
            fx = f8[0].float() if f8 is not None else ts.X.clone()
            fh = (f8[1].float() if f8 is not None else ts.Xp.clone().view(B, Tn, 88)) if hist_on else ts.w_true.clone()
            out.append((dict(eng.losses()), fx.view(B, Tn, 88), fh, ts.w_true.clone()))
        runs[staged] = (out, eng.P.get_weights())
    for a, b in zip(runs["1"][0], runs["0"][0]):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
    for k, v in runs["1"][1].items():
        assert np.array_equal(v, runs["0"][1][k]), k


@pytest.mark.parametrize("source", ["rows", "windows"])
def test_cl_vae_batch_assembly_inside_the_fused_step_equals_the_gather_launch(dev, monkeypatch, source):
    """cl_vae: the fused step kernel assembles its own mini-batch rows (clv_vae_fused_step(stage)) -- three launches per
    step.  Against the same steps with the gather launch (CLV_STAGE_IN_LABEL=0): the staged frames and labels, every loss and
    every parameter after three replayed steps, bit for bit; a batch that does not fill its last workgroup's 16 rows."""
    from clvae_amd.engine import VaeEngine
    from clvae_amd.trainer import TrainStep, DevWindows
    B, L, Cn, nb = 40, 4, 2, 3
    cfg = O.vae_config(latent_dim=L, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(32)
    p = {k: f32(v) for k, v in O.vae_init_params(cfg, seed=3).items()}
    n = nb * B
    keys = ft(np.eye(Cn)[rng.integers(0, Cn, n)], dev)
    if source == "rows":
        fr = (rng.random((n, 2, 88)) < 0.05)
        cur, hist, idx = u8(fr[:, 1], dev), u8(fr[:, 0], dev), None
    else:
        store = u8(rng.random((n + 50, 88)) < 0.05, dev)
        starts = torch.as_tensor(rng.permutation(n + 40)[:n].astype(np.int64), device=dev)
        cur, hist = DevWindows(store, starts, 1), DevWindows(store, starts, 0)
        idx = torch.as_tensor(rng.permutation(n).astype(np.int64), device=dev)
    runs = {}
    for staged in ("1", "0"):
        monkeypatch.setenv("CLV_STAGE_IN_LABEL", staged)
        eng = VaeEngine(cfg, B, dev)
        assert eng.fused
        eng.P.set_weights(p)
        ts = TrainStep(eng, seed=78)
        ts.bind_batches(cur, hist, keys, idx=idx, period=nb, stride=B)
        assert (ts._label_stage() is not None) == (staged == "1")
        out = []
        for it in range(3):
            ts.step()
            torch.cuda.synchronize()
            out.append((dict(eng.losses()), ts.X.clone(), ts.Xp.clone(), ts.w_true.clone()))
        runs[staged] = (out, eng.P.get_weights())
    for a, b in zip(runs["1"][0], runs["0"][0]):
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y)
    for k, v in runs["1"][1].items():
        assert np.array_equal(v, runs["0"][1][k]), k
