"""-m gpu: the frames of a training step as BYTES (round 6).

The frame store keeps piano-roll frames as uint8 (88 B per frame); until round 5 the staging launch of the
large-batch path widened every batch to float32 (185 MB written per step at configuration 5, read back by five launches).
Now that launch copies the bytes (`clv_gather_rows_multi*` with src_u8 == 2) and the kernels that read frames take them as
they are (x_u8 / y_u8 / CLV_FRAMES_U8 of include/clvae.h).  A byte IS its float value, so every kernel must give BIT FOR BIT
what it gives for the float copy of the same frames -- that is what these tests hold, kernel by kernel and for a whole step.
"""
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


def F(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def U8(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)


def frames(rng, shape, kind):
    """'notes': 0/1 piano-roll at the JSB density; 'bytes': any uint8 value (velocities)"""
    return (rng.random(shape) < 0.0443).astype(np.uint8) if kind == 'notes' else rng.integers(0, 256, shape).astype(np.uint8)


@pytest.mark.parametrize("Bn,nx,Nn,kind", [(1024, 22528, 88, 'notes'), (256, 11264, 88, 'notes'), (100, 132, 88, 'bytes'),
                                           (37, 96, 20, 'notes'), (1, 4, 4, 'bytes'), (33, 200, 96, 'bytes')])
def test_dense_outer_reads_bytes(dev, Bn, nx, Nn, kind):
    from clvae_amd import ops
    rng = np.random.default_rng(Bn + nx)
    X = frames(rng, (Bn, nx), kind)
    G = (rng.standard_normal((Bn, Nn)) * np.exp(rng.standard_normal((Bn, 1)))).astype(np.float32)
    H = np.maximum(rng.standard_normal((Bn, Nn)), 0).astype(np.float32)
    hb = rng.standard_normal(Nn).astype(np.float32)
    outs = []
    for Xd in (F(X, dev), U8(X, dev)):
        out = torch.full((nx, Nn), -3.0, dtype=torch.float32, device=dev)
        cs, gd = torch.full((Nn,), -3.0, device=dev), torch.full((Nn,), -3.0, device=dev)
        ops.dense_outer_bf16(Bn, nx, Nn, Xd, nx, F(G, dev), Nn, out, colsum=cs, gdot=(F(H, dev), Nn, F(hb, dev), gd))
        outs.append((out, cs, gd))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    ref = X.astype(np.float64).T @ G.astype(np.float64)
    mag = np.abs(X.astype(np.float64)).T @ np.abs(G.astype(np.float64)) + 1e-30
    assert (np.abs(outs[1][0].cpu().numpy() - ref) / mag).max() < 2e-6


@pytest.mark.parametrize("Bn,nx,Nn,kind", [(1024, 22528, 88, 'notes'), (256, 11264, 88, 'notes'), (100, 136, 88, 'bytes'),
                                           (37, 96, 20, 'notes'), (1, 8, 4, 'bytes'), (70, 2000, 96, 'bytes')])
def test_dense_window_fwd_reads_bytes(dev, Bn, nx, Nn, kind):
    from clvae_amd import ops
    rng = np.random.default_rng(Bn + nx + 1)
    X = frames(rng, (Bn, nx), kind)
    K = (rng.standard_normal((nx, Nn)) * np.exp(rng.standard_normal((nx, 1)))).astype(np.float32)
    got = []
    for Xd in (F(X, dev), U8(X, dev)):
        ws = ops.Workspace(dev)
        buf, splits = ops.dense_window_fwd_bf16(Bn, nx, Nn, Xd, nx, F(K, dev), Nn, ws)
        torch.cuda.synchronize()
        got.append(buf.view(torch.float32)[:splits * Bn * Nn].clone())
    assert torch.equal(got[0], got[1])
    ref = X.astype(np.float64) @ K.astype(np.float64)
    mag = np.abs(X.astype(np.float64)) @ np.abs(K.astype(np.float64)) + 1e-30
    assert (np.abs(got[1].reshape(-1, Bn, Nn).cpu().numpy().astype(np.float64).sum(0) - ref) / mag).max() < 2e-6


@pytest.mark.parametrize("R,defer,store", [(1, False, True), (37, False, True), (1000, True, False), (128 * 40 + 5, True, True)])
def test_out_head_reads_byte_targets(dev, R, defer, store):
    from clvae_amd import ops
    rng = np.random.default_rng(R)
    H = D = 88
    hs = np.tanh(rng.standard_normal((R, H))).astype(np.float32)
    Wo = (rng.standard_normal((H, D)) * 0.4).astype(np.float32)
    Wo[:, 3] *= 30.0
    bo = rng.standard_normal(D).astype(np.float32)
    Y = frames(rng, (R, D), 'notes')
    res = []
    for Yd in (F(Y, dev), U8(Y, dev)):
        z = lambda *sh: torch.full(sh, -7.0, dtype=torch.float32, device=dev)
        logits, dl = (z(R, D), z(R, D)) if store else (None, None)
        rownll, dhs, dWo, dbo = z(R), z(R, H), z(H, D), z(D)
        ws = ops.Workspace(dev)
        rq = ops.ReduceQueue(dev) if defer else None
        ops.out_head_train(R, H, D, F(hs, dev), F(Wo, dev), F(bo, dev), Yd, 1.0 / R, rownll, dhs, dWo, dbo, ws, logits=logits,
                           dlogits=dl, defer=rq)
        if rq is not None:
            rq.flush()
        torch.cuda.synchronize()
        res.append([rownll, dhs, dWo, dbo] + ([logits, dl] if store else []))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("K,Tn,nz,defer,nh", [(4096, 16, 0, True, 88), (4096, 16, 2, True, 88), (2304, 48, 32, False, 88),
                                              (1000, 8, 0, False, 88), (2048, 32, 8, True, 64), (1024, 16, 32, True, 96)])
def test_lstm_wgrad_reads_byte_frames(dev, K, Tn, nz, defer, nh):
    """clv_lstm_wgrad with x_exact_bf16 = CLV_FRAMES_U8: the frame rows as a uint8 [K, 88] batch, the z rows in their own
    buffer -- bit for bit the slabs of the float batch; ragged K (first / last stage paths), both kernel widths."""
    from clvae_amd import ops
    rng = np.random.default_rng(K + nz)
    N, nx = 352, 88
    X = frames(rng, (K, nx), 'notes')
    Z = rng.standard_normal((K, max(nz, 1) + 3)).astype(np.float32)
    hs = np.tanh(rng.standard_normal((K, nh))).astype(np.float32)
    dz = (rng.standard_normal((K, N)) * np.exp(rng.standard_normal((K, 1)) * 2)).astype(np.float32)
    res = []
    for Xd in (F(X, dev), U8(X, dev)):
        z = lambda *sh: torch.full(sh, -5.0, dtype=torch.float32, device=dev)
        dKx, dU, dKz = z(nx, N), z(nh, N), z(max(nz, 1), N)
        ws = ops.Workspace(dev)
        rq = ops.ReduceQueue(dev) if defer else None
        ops.lstm_wgrad(K, N, Xd, nx, nx, True, F(hs, dev), nh, nh, Tn, F(Z, dev) if nz else None, Z.shape[1], nz, F(dz, dev),
                       dKx, dU, dKz if nz else None, ws, defer=rq)
        if rq is not None:
            rq.flush()
        torch.cuda.synchronize()
        res.append((dKx, dU, dKz))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    ref = X.astype(np.float64).T @ dz.astype(np.float64)
    mag = np.abs(X.astype(np.float64)).T @ np.abs(dz.astype(np.float64)) + 1e-30
    assert (np.abs(res[1][0].cpu().numpy() - ref) / mag).max() < 4e-6


@pytest.mark.parametrize("B,Tn,nx,nz,kind,gate", [(7, 5, 88, 32, 'notes', 0), (1024, 3, 88, 32, 'notes', 0), (1030, 2, 88, 5, 'bytes', 1),
                                                  (4, 9, 88, 0, 'notes', 0), (6, 4, 96, 2, 'bytes', 0), (3, 12, 60, 0, 'notes', 1)])
def test_lstm_mx_fwd_reads_byte_frames(dev, B, Tn, nx, nz, kind, gate):
    """clv_lstm_mx_fwd with x_u8: note lists built from byte frames (row stride nx + 4 bytes) -- h, coef and aux bit for bit
    those of the float frames; frames with every note on, values up to 255."""
    from clvae_amd import ops
    rng = np.random.default_rng(B + Tn + nx)
    H, ldx, ldz = 88, nx + 4, nz + 3
    Xb = np.full((B * Tn, ldx), 9, np.uint8)
    Xb[:, :nx] = frames(rng, (B * Tn, nx), kind)
    Xb[1 % (B * Tn), :nx] = 1
    U = O.orthogonal(rng, (H, 4 * H), np.float64) * 1.5
    Kx = rng.standard_normal((nx, 4 * H)) * (0.7 if kind == 'notes' else 0.01)
    Kz = rng.standard_normal((max(nz, 1), 4 * H)) * 0.4
    Zb = rng.standard_normal((B * Tn, ldz))
    rb = rng.standard_normal((B, 4 * H)) * 0.3
    res = []
    for Xd in (F(Xb, dev), U8(Xb, dev)):
        hs = torch.full((B * Tn, H), 9.0, device=dev)
        coef = torch.full((B * Tn, 4 * H), 9.0, device=dev)
        aux = torch.full((B * Tn, 2 * H), 9.0, device=dev)
        ops.lstm_mx_fwd(B, Tn, Xd, ldx, nx, F(Kx, dev), F(Zb, dev) if nz else None, ldz, nz, F(Kz, dev) if nz else None,
                        F(rb, dev), F(U, dev), hs, coef, aux, gate_act=gate)
        torch.cuda.synchronize()
        res.append((hs, coef, aux))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("rows,n,row,chunk,ld,use_idx", [(40, 300, 88 * 4, 88, 96, True), (33, 90, 88, 0, 0, True),
                                                         (25, 100, 88 * 3, 88, 88, False), (9, 30, 60 * 2, 60, 60, True),
                                                         (1024, 2048, 88 * 256, 88, 88, True)])
def test_gather_copies_bytes(dev, rows, n, row, chunk, ld, use_idx):
    """src_u8 == 2: uint8 rows of the store land as bytes in a uint8 batch (chunked rows, windows of a frame store through a
    table); next to a float segment in the same launch."""
    from clvae_amd import ops
    rng = np.random.default_rng(rows)
    St = rng.integers(0, 256, (n, row)).astype(np.uint8)
    Wl = rng.standard_normal((n, 4)).astype(np.float32)
    idx = torch.as_tensor(rng.permutation(n)[:rows].astype(np.int64), device=dev) if use_idx else None
    row0 = 0 if use_idx else 7
    pieces = row // chunk if chunk else 1
    width = ld if chunk else row
    out = torch.full((rows * pieces, width), 77, dtype=torch.uint8, device=dev)
    wout = torch.zeros(rows, 4, dtype=torch.float32, device=dev)
    ops.gather_rows_multi(rows, idx, [(U8(St, dev), out, row, chunk, ld), (F(Wl, dev), wout, 4, 0, 0)], row0=row0)
    torch.cuda.synchronize()
    sel = idx.cpu().numpy() if use_idx else np.arange(row0, row0 + rows)
    got = out.cpu().numpy()
    w = chunk if chunk else row
    np.testing.assert_array_equal(got[:, :w], St[sel].reshape(rows * pieces, w))
    assert (got[:, w:] == 77).all()
    np.testing.assert_array_equal(wout.cpu().numpy(), Wl[sel])
    with pytest.raises(TypeError):
        ops.gather_rows_multi(rows, idx, [(F(St, dev), out, row, chunk, ld)], row0=row0)      # float source, uint8 output


@pytest.mark.parametrize("B,Tn,L,Cn", [(1024, 6, 32, 10), (768, 5, 12, 3), (64, 16, 2, 10), (256, 128, 2, 10), (7, 3, 8, 4)])
def test_step_on_byte_batch_equals_step_on_float_batch(dev, B, Tn, L, Cn):
    """One captured training step fed by the bound-batch cursor from a uint8 data set: with the byte batch (default) and with
    CLV_FRAMES_U8=0 (the float staging of rounds 4-5) -- the same losses, the same parameters after three steps, bit for bit
    (tests/test_gpu_timed_step.py holds the same step to the oracle).  The large-batch path (768+ rows: the staging launch
    copies the bytes) and the pair path (the label launch leaves the byte batch; BASELINE configuration 3 is the 256 x 128 case)."""
    import os
    from clvae_amd.engine import VrnnEngine
    from clvae_amd.trainer import TrainStep
    cfg = O.vrnn_config(latent_dim=L, seq_length=Tn, n_classes=Cn, use_x_prev=True)
    rng = np.random.default_rng(B)
    p0 = {k: np.asarray(v, np.float32) for k, v in O.vrnn_init_params(cfg, seed=3).items()}
    n = 2 * B
    win = frames(rng, (n, Tn + 1, 88), 'notes')
    cur, hist = U8(win[:, 1:], dev), U8(win[:, :-1], dev)
    wl = np.eye(Cn, dtype=np.float32)[rng.integers(0, Cn, n)]
    perm = torch.as_tensor(rng.permutation(n).astype(np.int64), device=dev)
    runs = []
    for flag in ('1', '0'):
        os.environ['CLV_FRAMES_U8'] = flag
        try:
            eng = VrnnEngine(cfg, B, dev)
            assert (eng.use_mx or eng.fuse_pair) and eng.frames_u8_supported() == (flag == '1')
            assert eng.frames_u8_route() == ((('gather' if eng.use_mx else 'label')) if flag == '1' else None)
            eng.P.set_weights(p0)
            ts = TrainStep(eng, seed=5, use_graph=True)
            ts.bind_batches(cur, hist, F(wl, dev), idx=perm, period=2, stride=B)
            losses = []
            for _ in range(3):
                ts.step()
                torch.cuda.synchronize()
                losses.append(dict(eng.losses()))
            assert (ts._f8 is not None) == (flag == '1')
            runs.append((losses, {k: v.copy() for k, v in eng.P.get_weights().items()}))
        finally:
            os.environ.pop('CLV_FRAMES_U8', None)
    for la, lb in zip(runs[0][0], runs[1][0]):
        assert la == lb, (la, lb)
    for k in runs[0][1]:
        np.testing.assert_array_equal(runs[0][1][k], runs[1][1][k], err_msg=k)


@pytest.mark.parametrize("R,nx,ldx,two", [(1, 88, 88, False), (1000, 88, 92, True), (33, 96, 96, True), (515, 16, 16, False)])
def test_sparse_proj_reads_byte_frames(dev, R, nx, ldx, two):
    """clv_sparse_proj / clv_sparse_proj2 with x_u8: the row gathers of byte frames (any value, padded rows) bit for bit those of
    the float frames."""
    from clvae_amd import ops
    rng = np.random.default_rng(R + nx)
    Nn = 352
    X = np.full((R, ldx), 7, np.uint8)
    X[:, :nx] = frames(rng, (R, nx), 'notes' if two else 'bytes')
    Ka, Kb = rng.standard_normal((nx, Nn)).astype(np.float32), rng.standard_normal((nx, Nn)).astype(np.float32)
    res = []
    for Xd in (F(X, dev), U8(X, dev)):
        oa, ob = torch.full((R, Nn), -1.0, device=dev), torch.full((R, Nn), -1.0, device=dev)
        if two:
            ops.sparse_proj2(R, Nn, (nx, Xd, ldx, F(Ka, dev), oa), (nx, Xd, ldx, F(Kb, dev), ob))
        else:
            ops.sparse_proj(R, nx, Nn, Xd, ldx, F(Ka, dev), oa)
        torch.cuda.synchronize()
        res.append((oa, ob))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[1][0].cpu().numpy(), X[:, :nx].astype(np.float64) @ Ka.astype(np.float64), rtol=1e-5, atol=1e-3)
