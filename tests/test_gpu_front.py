"""-m gpu: the label launch that also forms the LSTMs' frame projections (csrc/label_head.hip: vrnn_front_kernel, round 6)
against the two launches it replaces (clv_vrnn_label_fwd_x + clv_sparse_proj2; CLV_FRONT_FUSED=0): captured training steps
fed by the bound-batch cursor, every loss and every parameter after three steps BIT FOR BIT.  The oracle holds the same steps
in tests/test_gpu_timed_step.py and tests/test_gpu_switches.py."""
import os

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


def u8(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)


# B, T, L, C, history frames, where the rows come from, note density
CASES = [(256, 128, 2, 10, True, "rows", 0.05),          # BASELINE configuration 3: one batch row per projection workgroup and group
         (64, 16, 2, 10, True, "windows", 0.05),         # the reference's default seq_length, windows of one frame store
         (7, 3, 8, 4, True, "rows", 0.05),               # fewer rows than workgroups, fewer frames than waves
         (300, 5, 2, 4, True, "windows", 0.05),          # several rows per projection workgroup (5 of 300 over 64), ragged
         (17, 1, 2, 3, True, "rows", 0.3),               # one frame per row
         (40, 9, 2, 5, False, "windows", 0.05),          # a decoder without history frames: one projection
         (33, 20, 4, 6, True, "rows", 0.6)]              # dense frames: both halves of the 88 inputs, dozens of notes per frame


@pytest.mark.parametrize("B,Tn,L,Cn,hist_on,source,density", CASES)
def test_front_launch_equals_label_launch_plus_projection_launch(dev, B, Tn, L, Cn, hist_on, source, density):
    from clvae_amd import ops
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep, DevWindows
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=hist_on)
    rng = np.random.default_rng(B + Tn)
    p0 = {k: np.asarray(v, np.float32) for k, v in O.vrnn_init_params(cfg, seed=4).items()}
    nb = 2
    n = nb * B
    keys = torch.as_tensor(np.eye(Cn, dtype=np.float32)[rng.integers(0, Cn, n)], device=dev)
    if source == "rows":
        win = rng.random((n, Tn + 1, 88)) < density
        cur, hist = u8(win[:, 1:].reshape(n, -1), dev), (u8(win[:, :-1].reshape(n, -1), dev) if hist_on else None)
    else:
        store = u8(rng.random((n + Tn + 40, 88)) < density, dev)
        starts = torch.as_tensor(rng.permutation(n + 30)[:n].astype(np.int64), device=dev)
        cur, hist = DevWindows(store, starts, 1), (DevWindows(store, starts, 0) if hist_on else None)
    idx = torch.as_tensor(rng.permutation(n).astype(np.int64), device=dev)
    assert ops.vrnn_label_fwd_x_proj_supported(B, 88, Tn * 88, Tn, 352)
    runs = []
    for flag in ('1', '0'):
        os.environ['CLV_FRONT_FUSED'] = flag
        try:
            eng = VrnnEngine(cfg, B, dev)
            assert eng.fuse_pair and eng.front_fused == (flag == '1') and eng.frames_u8_route() == 'label'
            eng.P.set_weights(p0)
            ts = TrainStep(eng, seed=11, use_graph=True)
            ts.bind_batches(cur, hist, keys, idx=idx, period=nb, stride=B)
            assert ts._label_stage() is not None
            losses = []
            for _ in range(3):
                ts.step()
                torch.cuda.synchronize()
                losses.append(dict(eng.losses()))
            assert all(np.isfinite(v) for d in losses for v in d.values())
            runs.append((losses, {k: v.copy() for k, v in eng.P.get_weights().items()}, ts._f8[0].clone()))
        finally:
            os.environ.pop('CLV_FRONT_FUSED', None)
    for la, lb in zip(runs[0][0], runs[1][0]):
        assert la == lb, (la, lb)
    assert torch.equal(runs[0][2], runs[1][2])
    for k in runs[0][1]:
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k], err_msg=k)


def test_projection_rider_needs_the_stage_it_reads_through(dev):
    """clv_vrnn_label_fwd_x(proj) without a stage (or with a stage that leaves no byte batch) is refused, not run on garbage."""
    from clvae_amd import ops
    assert not ops.vrnn_label_fwd_x_proj_supported(4, 88, 88 * 3 + 1, 3, 352)      # nx is not T frames
    assert not ops.vrnn_label_fwd_x_proj_supported(4, 88, 88 * 3, 3, 800)          # half a kernel does not fit 3 columns per lane
    z = torch.zeros(1024, device=dev)
    with pytest.raises(ValueError):
        ops.vrnn_label_fwd_x(4, 88, 3, 352, z, 88, 88, z, z, z, z, z, z, z, 0.0, z, z, z, z, z, z, z, z, z,
                             proj=(1, 352, z, z, None, None))


def test_front_launch_soak_200_steps_bit_for_bit(dev):
    """BASELINE configuration 3's shape, 200 replayed steps over 8 different mini-batches: the merged launch (two kinds of workgroup
    sharing every CU, frames requested four ahead, buffer stores) against the two launches -- every parameter bit for bit at the
    end (a race or a stale prefetch would not survive 200 steps of an optimizer that amplifies a one-ulp difference)."""
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep
    B, Tn, L, Cn, nb, steps = 256, 128, 2, 10, 8, 200
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(77)
    p0 = {k: np.asarray(v, np.float32) for k, v in O.vrnn_init_params(cfg, seed=8).items()}
    n = nb * B
    win = rng.random((n, Tn + 1, 88)) < 0.0443
    cur, hist = u8(win[:, 1:].reshape(n, -1), dev), u8(win[:, :-1].reshape(n, -1), dev)
    keys = torch.as_tensor(np.eye(Cn, dtype=np.float32)[rng.integers(0, Cn, n)], device=dev)
    idx = torch.as_tensor(rng.permutation(n).astype(np.int64), device=dev)
    out = []
    for flag in ('1', '0'):
        os.environ['CLV_FRONT_FUSED'] = flag
        try:
            eng = VrnnEngine(cfg, B, dev)
            eng.P.set_weights(p0)
            ts = TrainStep(eng, seed=21, use_graph=True)
            ts.bind_batches(cur, hist, keys, idx=idx, period=nb, stride=B)
            for _ in range(steps):
                ts.step()
            torch.cuda.synchronize()
            out.append((dict(eng.losses()), {k: v.copy() for k, v in eng.P.get_weights().items()}))
        finally:
            os.environ.pop('CLV_FRONT_FUSED', None)
    assert all(np.isfinite(v) for v in out[0][0].values())
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for k in out[0][1]:
        np.testing.assert_array_equal(out[0][1][k], out[1][1][k], err_msg=k)
