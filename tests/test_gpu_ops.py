"""-m gpu: every HIP op of include/clvae.h against the numpy oracle, through the C ABI."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import clvae_oracle as O
from oracle import philox as OP

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import clvae_amd
    from clvae_amd import _lib
    _lib.require_gpu()          # fail loudly: no CPU fallback
    return torch.device("cuda:0")


def T(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def N(t):
    return t.detach().cpu().numpy().astype(np.float64)


GEMM_CASES = [
    # M, N, K, ta, tb, bias, act, beta, split
    (37, 29, 53, 0, 0, 1, 0, 0.0, 1),
    (64, 64, 16, 0, 0, 0, 1, 0.0, 1),
    (100, 88, 88, 0, 0, 1, 1, 0.0, 1),
    (100, 1, 88, 0, 0, 1, 0, 0.0, 1),        # w_mean head of cl_vae (N=1)
    (512, 352, 88, 0, 0, 0, 0, 0.0, 1),      # x-projection tile shape (64x176)
    (300, 352, 10, 0, 0, 1, 0, 1.0, 1),
    (256, 88, 1408, 0, 0, 1, 1, 0.0, 11),    # hW forward, split-K
    (88, 352, 2000, 1, 0, 0, 0, 0.0, 16),    # weight gradient (96x96 tiles), split-K
    (88, 352, 777, 1, 0, 0, 0, 1.0, 5),
    (1408, 88, 40, 1, 0, 0, 0, 0.0, 1),      # hW weight gradient
    (200, 88, 352, 0, 1, 0, 3, 0.0, 1),      # dX with relu mask
    (130, 2, 352, 0, 1, 0, 0, 0.0, 1),       # dZ (N=L=2)
    (2, 352, 600, 1, 0, 0, 0, 0.0, 4),       # Z rows of the decoder kernel gradient
    (45, 70, 33, 1, 1, 1, 2, 0.0, 1),
    (70, 130, 9, 0, 0, 1, 0, 0.5, 1),
]


@pytest.mark.parametrize("M,Nn,K,ta,tb,bias,act,beta,split", GEMM_CASES)
def test_gemm(dev, M, Nn, K, ta, tb, bias, act, beta, split):
    from clvae_amd import ops
    rng = np.random.default_rng(M * 131 + Nn * 17 + K)
    A = rng.standard_normal((K, M) if ta else (M, K))
    Bm = rng.standard_normal((Nn, K) if tb else (K, Nn))
    C0 = rng.standard_normal((M, Nn))
    b = rng.standard_normal(Nn) if bias else None
    aux = rng.standard_normal((M, Nn))
    ref = (A.T if ta else A).astype(np.float32).astype(np.float64) @ (Bm.T if tb else Bm).astype(np.float32).astype(np.float64)
    ref = 0.75 * ref
    if bias:
        ref = ref + b.astype(np.float32)
    ref = ref + beta * C0.astype(np.float32)
    if act == 1:
        ref = np.maximum(ref, 0)
    elif act == 2:
        ref = 1 / (1 + np.exp(-ref))
    elif act == 3:
        ref = ref * (aux.astype(np.float32) > 0)
    Cd = T(C0, dev)
    ws = ops.Workspace(dev)
    ops.gemm(T(A, dev), T(Bm, dev), Cd, M, Nn, K, ta=bool(ta), tb=bool(tb), alpha=0.75, beta=beta,
             bias=None if b is None else T(b, dev), act=act, aux=T(aux, dev) if act == 3 else None,
             split_k=split, ws=ws)
    torch.cuda.synchronize()
    scale = np.abs(ref).max() + 1
    assert np.abs(N(Cd) - ref).max() / scale < 2e-6 * max(1, np.sqrt(K) / 4)


def test_gemm_strided_views(dev):
    """ldc / lda column-block views used for the [mean|log_var] heads."""
    from clvae_amd import ops
    rng = np.random.default_rng(0)
    M, K, L = 50, 88, 4
    A = rng.standard_normal((M, K)); W1 = rng.standard_normal((K, L)); W2 = rng.standard_normal((K, L))
    out = torch.zeros(M, 2 * L, device=dev)
    ws = ops.Workspace(dev)
    ops.gemm(T(A, dev), T(W1, dev), out, M, L, K, ldc=2 * L, ws=ws)
    ops.gemm(T(A, dev), T(W2, dev), out[:, L:], M, L, K, ldc=2 * L, ws=ws)
    ref = np.concatenate([A.astype(np.float32) @ W1.astype(np.float32), A.astype(np.float32) @ W2.astype(np.float32)], 1)
    np.testing.assert_allclose(N(out), ref, atol=2e-5)
    back = torch.zeros(M, K, device=dev)
    ops.gemm(out, T(W1, dev), back, M, K, L, tb=True, lda=2 * L, ws=ws)
    np.testing.assert_allclose(N(back), ref[:, :L] @ W1.astype(np.float32).T, atol=1e-4)


def test_colsum_and_sum(dev):
    from clvae_amd import ops
    rng = np.random.default_rng(1)
    X = rng.standard_normal((3000, 90))
    out = torch.zeros(90, device=dev)
    ws = ops.Workspace(dev)
    ops.colsum(T(X, dev), 3000, 90, out, ws)
    np.testing.assert_allclose(N(out), X.astype(np.float32).astype(np.float64).sum(0), atol=2e-3)
    out2 = torch.ones(4, device=dev)
    ops.colsum(T(X, dev)[:, 5:], 3000, 4, out2, ws, ldx=90, beta=1.0)
    np.testing.assert_allclose(N(out2), 1 + X[:, 5:9].astype(np.float32).astype(np.float64).sum(0), atol=2e-3)
    s = torch.zeros(2, device=dev)
    ops.sum_strided(3000, T(X, dev)[:, 2:], 90, 1.0 / 3000, s)
    assert abs(N(s)[0] - X[:, 2].astype(np.float32).mean()) < 1e-5


@pytest.mark.parametrize("B,Tn,gate", [(3, 9, 0), (5, 1, 0), (4, 17, 1), (258, 4, 0), (516, 3, 1), (1024, 3, 0), (2048, 2, 0)])
def test_lstm_seq_fwd_bwd(dev, B, Tn, gate):
    from clvae_amd import ops
    H = 88
    rng = np.random.default_rng(B + Tn)
    U = O.orthogonal(rng, (H, 4 * H), np.float64) * 1.5
    xproj = rng.standard_normal((B, Tn, 4 * H)) * 1.5
    rb = rng.standard_normal((B, 4 * H)) * 0.3
    h0 = rng.standard_normal((B, H)) * 0.5
    c0 = rng.standard_normal((B, H)) * 0.5
    act = 'hard_sigmoid' if gate == 0 else 'sigmoid'
    # oracle: identity input kernel so that xs == xproj + rowbias
    xs = xproj + rb[:, None, :]
    hs_ref, cache = O.lstm_forward(xs, np.eye(4 * H), U, np.zeros(4 * H), h0=h0, c0=c0, gate_act=act)
    dHs = rng.standard_normal((B, Tn, H))
    _, _, _, _, dZ_ref = O.lstm_backward(dHs, cache, np.eye(4 * H), U)

    gates = T(xproj, dev)
    hs = torch.empty(B, Tn, H, device=dev); cs = torch.empty(B, Tn, H, device=dev)
    hT = torch.empty(B, H, device=dev); cT = torch.empty(B, H, device=dev)
    Ud = T(U, dev)
    ops.lstm_seq_fwd(B, Tn, gates, T(rb, dev), Ud, hs, cs, gates, h0=T(h0, dev), c0=T(c0, dev), hT=hT, cT=cT,
                     gate_act=gate)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(hs), hs_ref, atol=3e-6)
    np.testing.assert_allclose(N(cs), cache['C'], atol=5e-6)
    np.testing.assert_allclose(N(hT), hs_ref[:, -1], atol=3e-6)
    np.testing.assert_allclose(N(cT), cache['C'][:, -1], atol=5e-6)
    g = N(gates).reshape(B, Tn, 4, H)
    Zr = cache['Z'].reshape(B, Tn, 4, H)
    np.testing.assert_allclose(g[:, :, 0], Zr[:, :, 0], atol=1e-5)
    np.testing.assert_allclose(g[:, :, 2], np.tanh(Zr[:, :, 2]), atol=3e-6)
    dzsum = torch.empty(B, 4 * H, device=dev)
    ops.lstm_seq_bwd(B, Tn, Ud, T(dHs, dev), cs, gates, dzsum, c0=T(c0, dev), gate_act=gate)
    torch.cuda.synchronize()
    dz = N(gates).reshape(B, Tn, 4 * H)
    # hard-sigmoid kinks: a pre-activation within fp32 noise of +-2.5 may fall on the other side
    bad = np.abs(dz - dZ_ref) > 2e-5 * (1 + np.abs(dZ_ref))
    assert bad.mean() < 1e-4, bad.mean()
    np.testing.assert_allclose(N(dzsum), dz.sum(1), atol=1e-4)


@pytest.mark.parametrize("H,B,Tn,gate", [(1, 3, 4, 0), (7, 2, 5, 1), (32, 5, 9, 0), (50, 4, 3, 1), (64, 3, 16, 0), (100, 2, 6, 0),
                                         (128, 3, 8, 1), (257, 2, 3, 0), (600, 1, 2, 0), (1024, 1, 2, 1), (88, 4, 6, 0)])
def test_lstm_seq_fwd_bwd_at_any_width(dev, monkeypatch, H, B, Tn, gate):
    """clv_lstm_seq_fwd / _bwd for H != 88 (csrc/lstm_any.hip; H == 88: the same kernels forced by CLV_LSTM_ANY=1): states,
    stored gates, final state, dz and its column sums against the oracle's LSTM (cl_vrnn/model.py:196-199 under K.rnn /
    K.gradients); 1, 2, 4, 8 k-slices per unit, widths beyond the 256 threads (two and four units per owner thread), an
    initial state; then the stateful single-step form of sampling (T = 1, no cell / gate buffers, state in and out)."""
    from clvae_amd import ops
    if H == 88:
        monkeypatch.setenv("CLV_LSTM_ANY", "1")
    rng = np.random.default_rng(H * 31 + Tn)
    U = (O.orthogonal(rng, (H, 4 * H), np.float64) if H > 1 else rng.standard_normal((1, 4))) * 1.5
    xproj = rng.standard_normal((B, Tn, 4 * H)) * 1.5
    rb = rng.standard_normal((B, 4 * H)) * 0.3
    h0, c0 = rng.standard_normal((B, H)) * 0.5, rng.standard_normal((B, H)) * 0.5
    act = 'hard_sigmoid' if gate == 0 else 'sigmoid'
    hs_ref, cache = O.lstm_forward(xproj + rb[:, None, :], np.eye(4 * H), U, np.zeros(4 * H), h0=h0, c0=c0, gate_act=act)
    dHs = rng.standard_normal((B, Tn, H))
    _, _, _, _, dZ_ref = O.lstm_backward(dHs, cache, np.eye(4 * H), U)
    gates = T(xproj, dev)
    hs = torch.empty(B, Tn, H, device=dev); cs = torch.empty(B, Tn, H, device=dev)
    hT = torch.empty(B, H, device=dev); cT = torch.empty(B, H, device=dev)
    Ud = T(U, dev)
    ops.lstm_seq_fwd(B, Tn, gates, T(rb, dev), Ud, hs, cs, gates, h0=T(h0, dev), c0=T(c0, dev), hT=hT, cT=cT,
                     gate_act=gate, H=H)
    torch.cuda.synchronize()
    tol = 3e-6 * max(1.0, np.sqrt(H / 88))
    np.testing.assert_allclose(N(hs), hs_ref, atol=tol)
    np.testing.assert_allclose(N(cs), cache['C'], atol=2 * tol)
    np.testing.assert_allclose(N(hT), hs_ref[:, -1], atol=tol)
    np.testing.assert_allclose(N(cT), cache['C'][:, -1], atol=2 * tol)
    g = N(gates).reshape(B, Tn, 4, H)
    Zr = cache['Z'].reshape(B, Tn, 4, H)
    for k in (0, 1, 3):
        np.testing.assert_allclose(g[:, :, k], Zr[:, :, k], atol=4 * tol)
    np.testing.assert_allclose(g[:, :, 2], np.tanh(Zr[:, :, 2]), atol=tol)
    dzsum = torch.empty(B, 4 * H, device=dev)
    ops.lstm_seq_bwd(B, Tn, Ud, T(dHs, dev), cs, gates, dzsum, c0=T(c0, dev), gate_act=gate, H=H)
    torch.cuda.synchronize()
    dz = N(gates).reshape(B, Tn, 4 * H)
    bad = np.abs(dz - dZ_ref) > 2e-5 * (1 + np.abs(dZ_ref)) * max(1.0, np.sqrt(H / 88))     # hard-sigmoid kinks: see above
    assert bad.mean() < 2e-4, bad.mean()
    np.testing.assert_allclose(N(dzsum), dz.sum(1), atol=1e-4)
    # one stateful step (cl_vrnn/model.py:122-125: the batch-1 step models of sample.py): h, c in and out, nothing else stored
    st_h, st_c = T(h0, dev), T(c0, dev)
    out = torch.empty(B, H, device=dev)
    ops.lstm_seq_fwd(B, 1, T(xproj[:, 0] + rb, dev), None, Ud, out, None, None, h0=st_h, c0=st_c, hT=st_h, cT=st_c, gate_act=gate, H=H)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(out), hs_ref[:, 0], atol=tol)
    np.testing.assert_allclose(N(st_h), hs_ref[:, 0], atol=tol)
    np.testing.assert_allclose(N(st_c), cache['C'][:, 0], atol=2 * tol)


@pytest.mark.parametrize("B,Tn,gate", [(1027, 7, 0), (768, 3, 1), (2050, 2, 0), (769, 1, 0)])
def test_lstm_seq_fwd_at_large_batches(dev, B, Tn, gate):
    """clv_lstm_seq_fwd at batches far beyond one row per CU (the generic chain's forward: dropout, CLV_USE_MX=0): 1, 2 or 4
    rows per workgroup by what divides the batch."""
    from clvae_amd import ops
    H = 88
    rng = np.random.default_rng(B + Tn)
    U = O.orthogonal(rng, (H, 4 * H), np.float64) * 1.5
    xproj = rng.standard_normal((B, Tn, 4 * H)) * 1.5
    rb = rng.standard_normal((B, 4 * H)) * 0.3
    act = 'hard_sigmoid' if gate == 0 else 'sigmoid'
    hs_ref, cache = O.lstm_forward(xproj + rb[:, None, :], np.eye(4 * H), U, np.zeros(4 * H), gate_act=act)
    gates = T(xproj, dev)
    hs = torch.empty(B, Tn, H, device=dev); cs = torch.empty(B, Tn, H, device=dev)
    hT = torch.empty(B, H, device=dev); cT = torch.empty(B, H, device=dev)
    ops.lstm_seq_fwd(B, Tn, gates, T(rb, dev), T(U, dev), hs, cs, gates, hT=hT, cT=cT, gate_act=gate)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(hs), hs_ref, atol=3e-6)
    np.testing.assert_allclose(N(cs), cache['C'], atol=5e-6)
    np.testing.assert_allclose(N(hT), hs_ref[:, -1], atol=3e-6)
    np.testing.assert_allclose(N(cT), cache['C'][:, -1], atol=5e-6)
    g = N(gates).reshape(B, Tn, 4, H)
    Zr = cache['Z'].reshape(B, Tn, 4, H)
    for k in (0, 1, 3):
        np.testing.assert_allclose(g[:, :, k], Zr[:, :, k], atol=1e-5)
    np.testing.assert_allclose(g[:, :, 2], np.tanh(Zr[:, :, 2]), atol=3e-6)


@pytest.mark.parametrize("B,Tn,nz,gate", [(6, 9, 32, 0), (1028, 3, 32, 0), (514, 4, 5, 1), (3, 1, 40, 0)])
def test_lstm_seq_bwd_z_also_returns_the_latent_gradient(dev, B, Tn, nz, gate):
    """clv_lstm_seq_bwd_z == clv_lstm_seq_bwd, plus dZ = dz . Kz^T from the same launch."""
    from clvae_amd import ops
    H = 88
    rng = np.random.default_rng(B + nz)
    U = T(O.orthogonal(rng, (H, 4 * H), np.float64) * 1.5, dev)
    Kz = rng.standard_normal((nz, 4 * H)) * 0.4
    xproj = T(rng.standard_normal((B, Tn, 4 * H)), dev)
    dHs = T(rng.standard_normal((B, Tn, H)), dev)
    hs = torch.empty(B, Tn, H, device=dev); cs = torch.empty(B, Tn, H, device=dev)
    g1 = xproj.clone()
    ops.lstm_seq_fwd(B, Tn, g1, None, U, hs, cs, g1, gate_act=gate)
    g2 = g1.clone()
    s1 = torch.empty(B, 4 * H, device=dev); s2 = torch.empty(B, 4 * H, device=dev)
    ops.lstm_seq_bwd(B, Tn, U, dHs, cs, g1, s1, gate_act=gate)
    dZ = torch.full((B * Tn, nz + 3), 7.0, device=dev)
    ops.lstm_seq_bwd_z(B, Tn, U, dHs, cs, g2, s2, T(Kz, dev), nz, dZ, nz + 3, gate_act=gate)
    torch.cuda.synchronize()
    assert torch.equal(g1, g2) and torch.equal(s1, s2)
    ref = f32(N(g1).reshape(B * Tn, 4 * H)) @ f32(Kz).T
    np.testing.assert_allclose(N(dZ)[:, :nz], ref, atol=2e-5 * max(1.0, np.abs(ref).max()))
    assert float(dZ[:, nz:].min()) == 7.0 and float(dZ[:, nz:].max()) == 7.0


def test_label_gauss_bernoulli(dev):
    from clvae_amd import ops
    rng = np.random.default_rng(3)
    B, Cn, L, D = 37, 10, 3, 88
    C1 = Cn - 1
    wargs = rng.standard_normal((B, 2 * C1)) * 0.7
    eps = rng.standard_normal((B, C1))
    y = np.eye(Cn)[rng.integers(0, Cn, B)]
    prior = 0.3
    m, lv = wargs[:, :C1], wargs[:, C1:]
    w_ref = O.logistic_normal(m, lv, eps)
    klw, dm_kl, dlv_kl = O.kl_w_prior(m, lv, prior)
    wrec, dw_rec = O.cce_keras(w_ref, y, C1)
    wd = T(wargs, dev)
    w = torch.empty(B, Cn, device=dev); rl = torch.empty(B, 3, device=dev)
    ops.label_fwd(B, Cn, wd, wd[:, C1:], 2 * C1, T(eps, dev), T(y, dev), prior, w, rl)
    np.testing.assert_allclose(N(w), w_ref, atol=2e-6)
    np.testing.assert_allclose(N(rl)[:, 0], klw, rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(N(rl)[:, 1], wrec, rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(N(rl)[:, 2], (w_ref.argmax(1) == y.argmax(1)).astype(float))
    dw = rng.standard_normal((B, Cn))
    cw, wkl, inv = 0.7, 0.9, 1.0 / B
    ds, dlv_s = O.logistic_normal_bwd(w_ref, dw + cw * inv * dw_rec, lv, eps)
    dout = torch.empty(B, 2 * C1, device=dev)
    ops.label_bwd(B, Cn, wd, wd[:, C1:], 2 * C1, T(eps, dev), T(y, dev), w, T(dw, dev), prior, cw, wkl, inv,
                  dout, dout[:, C1:], 2 * C1)
    np.testing.assert_allclose(N(dout)[:, :C1], ds + wkl * inv * dm_kl, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(N(dout)[:, C1:], dlv_s + wkl * inv * dlv_kl, rtol=1e-4, atol=2e-6)

    R = 101
    za = rng.standard_normal((R, 2 * L)); ez = rng.standard_normal((R, L))
    z = torch.zeros(R, L + 2, device=dev); kl = torch.empty(R, device=dev)
    ops.gauss_fwd(R, L, T(za, dev), T(ez, dev), z[:, 2:], L + 2, kl)
    kl_ref, dzm, dzlv = O.kl_gauss(za[:, :L], za[:, L:])
    np.testing.assert_allclose(N(z)[:, 2:], za[:, :L] + np.exp(za[:, L:] / 2) * ez, atol=3e-6)
    np.testing.assert_allclose(N(kl), kl_ref, rtol=2e-5, atol=1e-5)
    dz = rng.standard_normal((R, L)); dza = torch.empty(R, 2 * L, device=dev)
    ops.gauss_bwd(R, L, T(za, dev), T(ez, dev), T(dz, dev), L, 0.25, dza)
    np.testing.assert_allclose(N(dza)[:, :L], dz + 0.25 * dzm, atol=3e-6)
    np.testing.assert_allclose(N(dza)[:, L:], dz * ez * 0.5 * np.exp(za[:, L:] / 2) + 0.25 * dzlv, atol=5e-6)

    a = rng.standard_normal((R, D)) * 6
    a[0, :5] = [20.0, -20.0, 16.2, -16.2, 0.0]              # beyond Keras' epsilon clip
    yy = (rng.random((R, D)) < 0.2).astype(np.float64)
    loss_ref, g_ref = O.bce_from_logits_keras(a.astype(np.float32).astype(np.float64), yy)
    nll = torch.empty(R, device=dev); dl = torch.empty(R, D, device=dev)
    ops.bernoulli_nll(R, D, T(a, dev), T(yy, dev), D, 0.5, nll, dl)
    np.testing.assert_allclose(N(nll), loss_ref, rtol=3e-6, atol=1e-4)
    np.testing.assert_allclose(N(dl), 0.5 * g_ref, atol=2e-6)


def test_bce_clip_points_are_those_of_float32_keras(dev):
    """keras.losses.binary_crossentropy in float32 (cl_vae/model.py:190-191, cl_vrnn/model.py:241-242) clips sigmoid(a) to
    [float32(1e-7), float32(1 - 1e-7)] = [1e-7, 1 - 2^-23]: on the logits the UPPER clip is log(2^23 - 1) = 15.942385, the
    lower one -16.118095.  Logits on both sides of both points, through every kernel that carries the loss: the
    stand-alone NLL, the GEMM epilogue and the fused output head."""
    from clvae_amd import ops
    assert abs(O.LOGIT_CLIP_HI - 15.942385) < 1e-6 and abs(O.LOGIT_CLIP_LO + 16.118095) < 1e-6
    pts = np.array([15.9, 15.94, 15.95, 16.0, 16.1, 16.2, 30.0, -15.95, -16.0, -16.1, -16.12, -16.2, -30.0, 0.0, 3.0, -3.0])
    D = 88
    a = np.zeros((2, D))
    a[0, :pts.size] = pts
    a[1, :pts.size] = pts
    y = np.zeros((2, D)); y[1] = 1.0                      # both targets at every point
    loss_ref, g_ref = O.bce_from_logits_keras(a, y)
    # the exact-arithmetic form differs exactly where the logit lies between the two upper clip points
    _, g_sym = O.bce_from_logits_keras(a, y, clip='exact')
    differs = (g_ref != g_sym)
    assert differs[:, :pts.size].any() and not differs[:, pts.size:].any()
    assert set(pts[differs[0, :pts.size]]) == {15.95, 16.0, 16.1}
    nll = torch.empty(2, device=dev); dl = torch.empty(2, D, device=dev)
    ops.bernoulli_nll(2, D, T(a, dev), T(y, dev), D, 1.0, nll, dl)
    np.testing.assert_allclose(N(nll), loss_ref, rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(N(dl), g_ref, atol=2e-7)
    # the same logits out of a product: hs = e_0 row selector, Wo row 0 = the points (bias 0)
    H = 88
    hs = np.zeros((2, H)); hs[:, 0] = 1.0
    Wo = np.zeros((H, D)); Wo[0, :pts.size] = pts
    bo = np.zeros(D)
    lg2 = torch.empty(2, D, device=dev); dl2 = torch.empty(2, D, device=dev); rn2 = torch.empty(2, device=dev)
    ops.gemm_bce(T(hs, dev), T(Wo, dev), T(bo, dev), T(y, dev), 1.0, lg2, dl2, rn2, 2, D, H)
    np.testing.assert_allclose(N(rn2), loss_ref, rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(N(dl2), g_ref, atol=2e-7)
    ws = ops.Workspace(dev)
    rn3 = torch.empty(2, device=dev); dhs = torch.empty(2, H, device=dev)
    dWo = torch.empty(H, D, device=dev); dbo = torch.empty(D, device=dev); dl3 = torch.empty(2, D, device=dev)
    ops.out_head_train(2, H, D, T(hs, dev), T(Wo, dev), T(bo, dev), T(y, dev), 1.0, rn3, dhs, dWo, dbo, ws, dlogits=dl3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(rn3), loss_ref, rtol=2e-6, atol=2e-5)
    np.testing.assert_allclose(N(dl3), g_ref, atol=2e-7)
    np.testing.assert_allclose(N(dbo), g_ref.sum(0), atol=4e-7)


@pytest.mark.parametrize("weightnorm", [True, False])
def test_adam_wn_three_steps(dev, weightnorm):
    from clvae_amd.engine import FlatParams
    rng = np.random.default_rng(7)
    shapes = [('a/kernel', (200, 88)), ('a/bias', (88,)), ('b/kernel', (88, 3)), ('b/bias', (3,)),
              ('c/kernel', (10, 352)), ('c/recurrent_kernel', (88, 352)), ('c/bias', (352,))]
    P = FlatParams(shapes, dev)
    p = {n: rng.standard_normal(s) * 0.3 for n, s in shapes}
    P.set_weights(p)
    st = O.adam_wn_init(p, weightnorm=weightnorm)
    for step in range(3):
        g = {n: rng.standard_normal(s) * (0.1 + step) for n, s in shapes}
        for n, _ in shapes:
            P.g(n).copy_(T(g[n], dev))
        P.adam_step(weightnorm=weightnorm)
        O.adam_wn_step(p, {k: v.astype(np.float32).astype(np.float64) for k, v in g.items()}, st)
        got = P.get_weights()
        for n, _ in shapes:
            np.testing.assert_allclose(got[n], p[n], rtol=3e-5, atol=3e-6, err_msg="%s step %d" % (n, step))
    assert int(P.iterations.item()) == 3
    if weightnorm:
        np.testing.assert_allclose(N(P.s)[P.col_offsets['a/kernel']:][:88], st['s']['a/kernel'], rtol=3e-5)


def test_rmsprop_three_steps(dev):
    """The 'rmsprop' optimizer string (cl_vae/train.py:83): CLV_OPT_RMSPROP vs the oracle's Keras RMSprop."""
    from clvae_amd.engine import FlatParams
    rng = np.random.default_rng(9)
    shapes = [('a/kernel', (200, 88)), ('a/bias', (88,)), ('c/recurrent_kernel', (88, 352))]
    P = FlatParams(shapes, dev)
    p = {n: rng.standard_normal(s) * 0.3 for n, s in shapes}
    P.set_weights(p)
    acc = {n: np.zeros(s) for n, s in shapes}
    for step in range(3):
        g = {n: rng.standard_normal(s) * (0.1 + step) for n, s in shapes}
        for n, _ in shapes:
            P.g(n).copy_(T(g[n], dev))
        P.adam_step(b2=0.9, weightnorm=2)
        O.rmsprop_step(p, {k: v.astype(np.float32).astype(np.float64) for k, v in g.items()}, acc)
        got = P.get_weights()
        for n, _ in shapes:
            np.testing.assert_allclose(got[n], p[n], rtol=3e-5, atol=3e-6, err_msg="%s step %d" % (n, step))
    assert int(P.iterations.item()) == 3


def test_philox_matches_oracle(dev):
    from clvae_amd import ops
    for first in (0, 5, 1027):
        n = 4099
        out = torch.empty(n, device=dev)
        ops.philox_uniform(out, n, seed=0x123456789ABC, step=3, stream_id=2, first_index=first)
        np.testing.assert_array_equal(N(out).astype(np.float32),
                                      OP.uniform(n, 0x123456789ABC, 3, 2, first))
        ops.philox_normal(out, n, seed=99, step=1, stream_id=0, first_index=first)
        np.testing.assert_allclose(N(out), OP.normal(n, 99, 1, 0, first), atol=2e-5)
    # step read from a device counter == step passed by value
    ctr = torch.tensor([7], dtype=torch.int32, device=dev)
    a = torch.empty(1000, device=dev); b = torch.empty(1000, device=dev)
    ops.philox_normal(a, 1000, seed=5, step=2, step_dev=ctr)
    ops.philox_normal(b, 1000, seed=5, step=9)
    assert torch.equal(a, b)
    big = torch.empty(1 << 20, device=dev)
    ops.philox_normal(big, 1 << 20, seed=1234)
    assert abs(big.mean().item()) < 5e-3 and abs(big.std().item() - 1) < 5e-3
    u = torch.rand(1000, device=dev); pp = torch.rand(1000, device=dev); x = torch.empty(1000, device=dev)
    ops.bernoulli_sample(1000, pp, u, x)
    assert torch.equal(x, (u <= pp).float())


def test_graph_capture_replay(dev):
    from clvae_amd import ops
    A = torch.randn(64, 32, device=dev); Bm = torch.randn(32, 48, device=dev); Cc = torch.zeros(64, 48, device=dev)
    ws = ops.Workspace(dev)
    ops.gemm(A, Bm, Cc, 64, 48, 32, ws=ws)
    torch.cuda.synchronize()
    with ops.Graph() as gr:
        ops.gemm(A, Bm, Cc, 64, 48, 32, beta=1.0, ws=ws)
    ref = Cc.clone()
    for _ in range(3):
        gr.launch()
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(Cc), 4 * N(ref), rtol=1e-5, atol=1e-5)


def test_gemm_grouped_tn(dev):
    """one launch: x^T.dz, h_{t-1}^T.dz (shift 1, zero at t == 0), z^T.dz, and the column sums (ones row)."""
    from clvae_amd import ops
    rng = np.random.default_rng(5)
    B, Tn, H, Nn = 6, 7, 88, 352
    K = B * Tn
    X = rng.standard_normal((K, 88)); hs = rng.standard_normal((K, H)); Z = rng.standard_normal((K, 3))
    dz = rng.standard_normal((K, Nn))
    hprev = hs.reshape(B, Tn, H).copy()
    hprev[:, 1:] = hprev[:, :-1]; hprev[:, 0] = 0
    hprev = hprev.reshape(K, H)
    ws = ops.Workspace(dev)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    for split in (1, 3, None):
        dK = torch.zeros(88 + 3, Nn, device=dev); dU = torch.zeros(H, Nn, device=dev); db = torch.ones(Nn, device=dev)
        ops.gemm_grouped_tn([dict(A=T(X, dev), lda=88, M=88, C=dK),
                             dict(A=T(hs, dev), lda=H, M=H, C=dU, shift=1, zero_period=Tn),
                             dict(A=T(Z, dev), lda=3, M=3, C=dK[88:]),
                             dict(A=None, M=1, C=db, ones=True)], Nn, K, T(dz, dev), ws, split_k=split)
        torch.cuda.synchronize()
        np.testing.assert_allclose(N(dK)[:88], f(X).T @ f(dz), atol=2e-4)
        np.testing.assert_allclose(N(dK)[88:], f(Z).T @ f(dz), atol=2e-4)
        np.testing.assert_allclose(N(dU), f(hprev).T @ f(dz), atol=2e-4)
        np.testing.assert_allclose(N(db), f(dz).sum(0), atol=2e-4)


def test_loss_sums(dev):
    from clvae_amd import ops
    rng = np.random.default_rng(6)
    a, b, c = rng.standard_normal(5000), rng.standard_normal(37), rng.standard_normal((29, 3))
    out = torch.zeros(8, device=dev)
    cd = T(c, dev)
    ops.loss_sums([(T(a, dev), 5000, 1), (T(b, dev), 37, 1), (cd, 29, 3), (cd[:, 1:], 29, 3), (cd[:, 2:], 29, 3)], out)
    ref = [a.astype(np.float32).mean(), b.astype(np.float32).mean()] + [c[:, j].astype(np.float32).mean() for j in range(3)]
    np.testing.assert_allclose(N(out)[:5], ref, atol=1e-5)


@pytest.mark.parametrize("R,nx,Nn,ldx,dense", [(1, 88, 352, 88, False), (7, 88, 352, 92, False), (1000, 88, 352, 88, False),
                                               (33, 90, 352, 92, True), (515, 16, 64, 16, True)])
def test_sparse_proj_matches_dense(dev, R, nx, Nn, ldx, dense):
    """out = X . K with only the nonzero inputs visited; exact for binary frames and for dense float frames."""
    from clvae_amd import ops
    rng = np.random.default_rng(R + nx)
    X = np.zeros((R, ldx), np.float32)
    if dense:
        X[:, :nx] = rng.standard_normal((R, nx)).astype(np.float32)
    else:
        X[:, :nx] = (rng.random((R, nx)) < 0.05).astype(np.float32)
    X[:, nx:] = 7.0                              # padding columns must be ignored
    K = rng.standard_normal((nx, Nn)).astype(np.float32)
    ldo = Nn + 4
    out = torch.full((R, ldo), -1.0, dtype=torch.float32, device=dev)
    assert ops.sparse_proj_supported(nx, Nn)
    ops.sparse_proj(R, nx, Nn, T(X, dev), ldx, T(K, dev), out, ldo)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = X[:, :nx].astype(np.float64) @ K.astype(np.float64)
    np.testing.assert_allclose(got[:, :Nn], ref, rtol=1e-5, atol=1e-5)
    assert np.all(got[:, Nn:] == -1.0)


@pytest.mark.parametrize("R,nx,Nn,dense,relu", [(5, 11264, 88, False, True), (256, 11264, 88, False, True),
                                                (64, 22528, 88, False, True),       # config 5: T = 256 windows
                                                (3, 200, 18, True, False), (2, 64, 128, True, True)])
def test_sparse_dense_matches_dense(dev, R, nx, Nn, dense, relu):
    from clvae_amd import ops, _lib
    rng = np.random.default_rng(R * 7 + Nn)
    X = rng.standard_normal((R, nx)).astype(np.float32) if dense else (rng.random((R, nx)) < 0.0443).astype(np.float32)
    K = (rng.standard_normal((nx, Nn)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(Nn).astype(np.float32)
    out = torch.zeros(R, Nn, dtype=torch.float32, device=dev)
    ops.sparse_dense(R, nx, Nn, T(X, dev), nx, T(K, dev), T(bias, dev), _lib.ACT_RELU if relu else _lib.ACT_NONE, out)
    torch.cuda.synchronize()
    ref = X.astype(np.float64) @ K.astype(np.float64) + bias
    if relu:
        ref = np.maximum(ref, 0.0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("Bn,nx,Nn,dense", [(256, 11264, 88, False), (100, 130, 88, False), (300, 70, 18, True),
                                            (1024, 22528, 88, False),               # config 5: T = 256 windows
                                            (1, 64, 2, True)])
def test_sparse_outer_matches_dense(dev, Bn, nx, Nn, dense):
    from clvae_amd import ops
    rng = np.random.default_rng(Bn + nx)
    X = rng.standard_normal((Bn, nx)).astype(np.float32) if dense else (rng.random((Bn, nx)) < 0.0443).astype(np.float32)
    G = rng.standard_normal((Bn, Nn)).astype(np.float32)
    out = torch.full((nx, Nn), -3.0, dtype=torch.float32, device=dev)
    cs = torch.full((Nn,), -3.0, dtype=torch.float32, device=dev)
    ops.sparse_outer(Bn, nx, Nn, T(X, dev), nx, T(G, dev), Nn, out, colsum=cs)
    torch.cuda.synchronize()
    ref = X.astype(np.float64).T @ G.astype(np.float64)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), G.astype(np.float64).sum(0), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("Bn,nx,Nn,kind", [(256, 11264, 88, 'notes'),                # config 3
                                           (1024, 22528, 88, 'notes'),               # config 5: T = 256 windows
                                           (100, 132, 88, 'bytes'),                  # any uint8 value is exact in bf16
                                           (37, 96, 20, 'notes'), (1, 4, 4, 'bytes'), (33, 200, 96, 'bytes')])
def test_dense_outer_bf16_matches_fp64(dev, Bn, nx, Nn, kind):
    """clv_dense_outer_bf16: dK = X^T G for byte-valued X on the bf16 matrix cores (X one exact piece, G three), the column
    sums of G and gdot = sum_b (H - hb) G, against fp64 numpy; ragged batch / input / column counts; and against the
    note-walking kernel it replaces (same bars)."""
    from clvae_amd import ops
    rng = np.random.default_rng(Bn + nx + Nn)
    X = (rng.random((Bn, nx)) < 0.0443).astype(np.float32) if kind == 'notes' else rng.integers(0, 256, (Bn, nx)).astype(np.float32)
    G = (rng.standard_normal((Bn, Nn)) * np.exp(rng.standard_normal((Bn, 1)))).astype(np.float32)
    H = np.maximum(rng.standard_normal((Bn, Nn)), 0).astype(np.float32)
    hb = rng.standard_normal(Nn).astype(np.float32)
    assert ops.dense_outer_bf16_supported(Bn, nx, Nn, nx, Nn)
    assert not ops.dense_outer_bf16_supported(Bn, nx, 100, nx, 100)              # more than 96 columns
    out = torch.full((nx, Nn), -3.0, dtype=torch.float32, device=dev)
    cs, gd = torch.full((Nn,), -3.0, device=dev), torch.full((Nn,), -3.0, device=dev)
    ops.dense_outer_bf16(Bn, nx, Nn, T(X, dev), nx, T(G, dev), Nn, out, colsum=cs, gdot=(T(H, dev), Nn, T(hb, dev), gd))
    torch.cuda.synchronize()
    f8 = lambda a: a.astype(np.float64)
    ref = f8(X).T @ f8(G)
    mag = np.abs(f8(X)).T @ np.abs(f8(G)) + 1e-30
    err = np.abs(out.cpu().numpy() - ref) / mag
    assert err.max() < 2e-6, err.max()                 # an fp32 accumulation of exact products over Bn terms
    np.testing.assert_allclose(cs.cpu().numpy(), f8(G).sum(0), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gd.cpu().numpy(), ((f8(H) - f8(hb)) * f8(G)).sum(0), rtol=2e-5, atol=1e-4)
    out2 = torch.full((nx, Nn), -3.0, dtype=torch.float32, device=dev)
    if Nn % 2 == 0 and Nn <= 128:
        ops.sparse_outer(Bn, nx, Nn, T(X, dev), nx, T(G, dev), Nn, out2)
        torch.cuda.synchronize()
        assert (np.abs(out2.cpu().numpy() - out.cpu().numpy()) / mag).max() < 4e-6


@pytest.mark.parametrize("Bn,nx,Nn,kind", [(256, 11264, 88, 'notes'), (1024, 22528, 88, 'notes'), (100, 136, 88, 'bytes'),
                                           (37, 96, 20, 'notes'), (1, 8, 4, 'bytes'), (70, 2000, 96, 'bytes')])
def test_dense_window_fwd_bf16_matches_fp64(dev, Bn, nx, Nn, kind):
    """clv_dense_window_fwd_bf16: the split-K partial sums of X . K for byte-valued X on the bf16 matrix cores (X one exact
    piece, K three); their sum against fp64 numpy; ragged row / input / column counts, chunk borders inside a stage."""
    from clvae_amd import ops
    rng = np.random.default_rng(Bn + nx + Nn + 1)
    X = (rng.random((Bn, nx)) < 0.0443).astype(np.float32) if kind == 'notes' else rng.integers(0, 256, (Bn, nx)).astype(np.float32)
    K = (rng.standard_normal((nx, Nn)) * np.exp(rng.standard_normal((nx, 1)))).astype(np.float32)
    assert ops.dense_window_fwd_bf16_supported(Bn, nx, Nn, nx, Nn)
    assert not ops.dense_window_fwd_bf16_supported(Bn, nx + 4, Nn, nx + 4, Nn)        # inputs in whole groups of 8
    ws = ops.Workspace(dev)
    buf, splits = ops.dense_window_fwd_bf16(Bn, nx, Nn, T(X, dev), nx, T(K, dev), Nn, ws)
    torch.cuda.synchronize()
    parts = buf.view(torch.float32)[:splits * Bn * Nn].reshape(splits, Bn, Nn).cpu().numpy().astype(np.float64)
    f8 = lambda a: a.astype(np.float64)
    ref = f8(X) @ f8(K)
    mag = np.abs(f8(X)) @ np.abs(f8(K)) + 1e-30
    err = np.abs(parts.sum(0) - ref) / mag
    assert err.max() < 2e-6, err.max()
    # every chunk holds the sum over ITS inputs only
    chunk = -(-nx // splits)
    chunk = (chunk + 31) // 32 * 32
    for c in (0, splits - 1):
        sl = slice(c * chunk, min(nx, (c + 1) * chunk))
        e = np.abs(parts[c] - f8(X[:, sl]) @ f8(K[sl])) / mag
        assert e.max() < 2e-6, (c, e.max())


@pytest.mark.parametrize("M,Nn,K", [(100, 88, 88), (37, 18, 40), (70, 130, 24)])
def test_gemm_bce_matches_separate_kernels(dev, M, Nn, K):
    """Output head with the NLL fused into the GEMM epilogue == GEMM followed by clv_bernoulli_nll."""
    from clvae_amd import ops
    rng = np.random.default_rng(M + Nn)
    A = (rng.standard_normal((M, K)) * 3).astype(np.float32)       # large logits: exercises the epsilon clip
    Bm = rng.standard_normal((K, Nn)).astype(np.float32)
    bias = rng.standard_normal(Nn).astype(np.float32)
    Y = (rng.random((M, Nn)) < 0.2).astype(np.float32)
    dA, dB, db, dY = T(A, dev), T(Bm, dev), T(bias, dev), T(Y, dev)
    z = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
    lg1, dl1, rn1, lg2, dl2, rn2 = z(M, Nn), z(M, Nn), z(M), z(M, Nn), z(M, Nn), z(M)
    ws = ops.Workspace(dev)
    ops.gemm(dA, dB, lg1, M, Nn, K, bias=db, ws=ws)
    ops.bernoulli_nll(M, Nn, lg1, dY, Nn, 0.37, rn1, dl1)
    ops.gemm_bce(dA, dB, db, dY, 0.37, lg2, dl2, rn2, M, Nn, K)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(lg2), N(lg1), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(N(dl2), N(dl1), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(N(rn2), N(rn1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,n,row,chunk,ld,use_idx", [(40, 300, 88 * 4, 88, 96, True), (33, 90, 88, 0, 0, True),
                                                         (16, 64, 87, 0, 0, True), (25, 100, 88 * 3, 0, 0, False)])
def test_gather_rows_multi_uint8_store(dev, rows, n, row, chunk, ld, use_idx):
    """Frames kept as uint8 in HBM gather to exactly the floats the float32 store gives (bit-exact), in one launch
    together with a float32 segment; aligned (4 bytes -> float4) and unaligned row lengths, chunked output rows."""
    from clvae_amd import ops
    rng = np.random.default_rng(rows)
    F = (rng.random((n, row)) < 0.1).astype(np.uint8)
    Wl = rng.standard_normal((n, 4)).astype(np.float32)
    idx = torch.as_tensor(rng.permutation(n)[:rows].astype(np.int64), device=dev) if use_idx else None
    row0 = 0 if use_idx else 7
    d_u8, d_f, d_w = torch.as_tensor(F, device=dev), T(F, dev), T(Wl, dev)
    pieces = row // chunk if chunk else 1
    width = ld if chunk else row
    outs = [torch.full((rows * pieces, width), -2.0, dtype=torch.float32, device=dev) for _ in range(2)]
    wout = [torch.zeros(rows, 4, dtype=torch.float32, device=dev) for _ in range(2)]
    ops.gather_rows_multi(rows, idx, [(d_u8, outs[0], row, chunk, ld), (d_w, wout[0], 4, 0, 0)], row0=row0)
    ops.gather_rows_multi(rows, idx, [(d_f, outs[1], row, chunk, ld), (d_w, wout[1], 4, 0, 0)], row0=row0)
    torch.cuda.synchronize()
    sel = idx.cpu().numpy() if use_idx else np.arange(row0, row0 + rows)
    assert torch.equal(outs[0], outs[1]) and torch.equal(wout[0], wout[1])
    got = outs[0].cpu().numpy()
    if chunk:
        got = got[:, :chunk].reshape(rows, row)
        assert (outs[0].cpu().numpy()[:, chunk:] == -2.0).all()          # padding columns untouched
    np.testing.assert_array_equal(got, F[sel].astype(np.float32))
    np.testing.assert_array_equal(wout[0].cpu().numpy(), Wl[sel])


@pytest.mark.parametrize("rows,pieces,use_idx,dense", [(5, 7, False, 0.05), (33, 16, True, 0.05), (4, 128, True, 0.3), (3, 2, False, 1.0)])
def test_gather_note_lists(dev, rows, pieces, use_idx, dense):
    """clv_gather_rows_multi(notes_out): next to the float copy of the frames, every frame's NOTE LIST -- the indices of its
    nonzero bytes (any order), then CLV_NOTE_NONE up to the end of the 96-byte row; frames with no note, more than 8 notes
    and all 88 notes; windows of a frame store through a start table."""
    from clvae_amd import ops
    rng = np.random.default_rng(rows * 100 + pieces)
    D = 88
    nwin = rows + 3
    store = (rng.random((nwin + pieces + 2, D)) < dense).astype(np.uint8)
    store[1] = 0
    starts = torch.as_tensor(np.arange(nwin, dtype=np.int64), device=dev)
    d_store = torch.as_tensor(store, device=dev)
    idx = torch.as_tensor(rng.permutation(nwin)[:rows].astype(np.int64), device=dev) if use_idx else None
    out = torch.full((rows, pieces, D), -1.0, device=dev)
    notes = torch.zeros(rows * pieces, ops.NOTE_ROW, dtype=torch.uint8, device=dev)
    # window i = frames starts[i] + 1 .. + pieces of the store (stride one frame, offset one frame)
    ops.gather_rows_multi(rows, idx, [(d_store, out, pieces * D, D, D, D, D, starts)], row0=2, notes=[notes])
    torch.cuda.synchronize()
    sel = idx.cpu().numpy() if use_idx else np.arange(2, 2 + rows)
    want = np.stack([store[i + 1:i + 1 + pieces] for i in sel])
    np.testing.assert_array_equal(out.cpu().numpy(), want.astype(np.float32))
    got = notes.cpu().numpy().reshape(rows, pieces, ops.NOTE_ROW)
    for r in range(rows):
        for p in range(pieces):
            on = np.flatnonzero(want[r, p])
            row = got[r, p]
            assert sorted(row[:on.size].tolist()) == on.tolist(), (r, p)
            assert (row[on.size:] == ops.NOTE_NONE).all()


@pytest.mark.parametrize("R,defer,store", [(1, False, True), (37, False, True), (128, True, True), (1000, True, False),
                                           (4096, True, True), (128 * 300 + 5, True, False)])
def test_out_head_train_matches_numpy(dev, R, defer, store):
    """clv_out_head_train: logits, row NLL, dhs, dWo, dbo against fp64 numpy (Keras' clipped BCE, cl_vrnn/model.py:229-242);
    ragged row counts, one block per workgroup and the persistent multi-block case, immediate and deferred reduction."""
    from clvae_amd import ops
    rng = np.random.default_rng(R)
    H = D = 88
    hs = np.tanh(rng.standard_normal((R, H))).astype(np.float32)
    Wo = (rng.standard_normal((H, D)) * 0.4).astype(np.float32)
    Wo[:, 3] *= 30.0                                     # one output column far outside the epsilon clip
    bo = rng.standard_normal(D).astype(np.float32)
    Y = (rng.random((R, D)) < 0.1).astype(np.float32)
    scale = 1.0 / R
    z = lambda *sh: torch.full(sh, -7.0, dtype=torch.float32, device=dev)
    logits, dl = (z(R, D), z(R, D)) if store else (None, None)
    rownll, dhs, dWo, dbo = z(R), z(R, H), z(H, D), z(D)
    ws = ops.Workspace(dev)
    rq = ops.ReduceQueue(dev) if defer else None
    ops.out_head_train(R, H, D, T(hs, dev), T(Wo, dev), T(bo, dev), T(Y, dev), scale, rownll, dhs, dWo, dbo, ws,
                       logits=logits, dlogits=dl, defer=rq)
    if rq is not None:
        rq.flush()
    torch.cuda.synchronize()
    a = hs.astype(np.float64) @ Wo.astype(np.float64) + bo
    l = np.clip(a, O.LOGIT_CLIP_LO, O.LOGIT_CLIP_HI)          # the float32 Keras clip points (asymmetric)
    nll = (np.maximum(l, 0) + np.log1p(np.exp(-np.abs(l))) - l * Y).sum(1)
    dlr = scale * (1 / (1 + np.exp(-l)) - Y) * ((a >= O.LOGIT_CLIP_LO) & (a <= O.LOGIT_CLIP_HI))
    np.testing.assert_allclose(N(rownll), nll, rtol=2e-5, atol=2e-4)
    if store:
        np.testing.assert_allclose(N(logits), a, rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(N(dl), dlr, rtol=1e-4, atol=2e-7 * scale * 10)
    np.testing.assert_allclose(N(dhs), dlr @ Wo.astype(np.float64).T, rtol=1e-4, atol=3e-5 * scale * 100)
    np.testing.assert_allclose(N(dWo), hs.astype(np.float64).T @ dlr, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(N(dbo), dlr.sum(0), rtol=1e-4, atol=2e-5)
    if store:
        # The weight-gradient PRODUCT alone: hs^T . dl with the device's own dl (fp32; sigmoid(l) - y cancels in fp32 where it
        # must, like in Keras) taken as exact input -- 6 of 9 bf16 piece pairs per product (csrc/out_head_bf16.hip: what is
        # dropped is <= 2^-24 of a product) plus the fp32 accumulation over the R rows, per entry against the sum of the
        # magnitudes of its products.  A single bf16 piece would be off by 2^-9.
        dl_dev = N(dl)
        ref = hs.astype(np.float64).T @ dl_dev
        mag = np.abs(hs.astype(np.float64)).T @ np.abs(dl_dev) + 1e-300
        err = (np.abs(N(dWo) - ref) / mag).max()
        assert err <= 2.0 ** -23 + np.sqrt(R) * 2.0 ** -24, (err, R)


def test_sparse_proj2_matches_two_single_launches(dev):
    """Both LSTM input projections in one launch == two clv_sparse_proj launches (bit-exact), different nx / ldx."""
    from clvae_amd import ops
    rng = np.random.default_rng(11)
    R, N = 5000, 352
    Xa = (rng.random((R, 88)) < 0.05).astype(np.float32)
    Xb = np.zeros((R, 92), dtype=np.float32)
    Xb[:, :70] = (rng.random((R, 70)) < 0.08)
    Xb[3, 5] = 0.37
    Ka, Kb = rng.standard_normal((88, N)).astype(np.float32), rng.standard_normal((70, N)).astype(np.float32)
    dXa, dXb, dKa, dKb = T(Xa, dev), T(Xb, dev), T(Ka, dev), T(Kb, dev)
    o = [torch.full((R, N), -1.0, dtype=torch.float32, device=dev) for _ in range(4)]
    ops.sparse_proj2(R, N, (88, dXa, 88, dKa, o[0]), (70, dXb, 92, dKb, o[1]))
    ops.sparse_proj(R, 88, N, dXa, 88, dKa, o[2])
    ops.sparse_proj(R, 70, N, dXb, 92, dKb, o[3])
    torch.cuda.synchronize()
    assert torch.equal(o[0], o[2]) and torch.equal(o[1], o[3])
    np.testing.assert_allclose(o[1].cpu().numpy(), Xb[:, :70].astype(np.float64) @ Kb, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("K,Tn,nz,exact,defer,nh", [(4096, 128, 0, True, False, 88), (4096, 64, 2, True, True, 88),
                                                    (1000, 8, 8, False, False, 88), (333 * 3, 3, 2, False, True, 88),
                                                    (32768, 128, 2, True, True, 88), (8192, 256, 32, True, False, 88),
                                                    (96, 96, 0, False, False, 88),
                                                    # windows that start at row 16 of a 32-row stage (the fast loader's second case)
                                                    (4096, 16, 2, True, True, 88), (2304, 48, 2, False, False, 88),
                                                    # other hidden sizes: every slot table of the producers is exercised
                                                    (2048, 32, 32, True, False, 64),      # nh + nz = 96 but nz > 8: wide kernel
                                                    (2048, 32, 8, False, True, 64),       # 6-row-tile kernel, 8 latent columns
                                                    (1024, 16, 32, True, False, 96),      # nh + nz = 128: the largest image
                                                    (1024, 16, 0, False, False, 96)])
def test_lstm_wgrad_split_bf16_matches_fp64(dev, K, Tn, nz, exact, defer, nh):
    """clv_lstm_wgrad: dKx = X^T dz, dU = H'^T dz (h of the previous step, zero at window starts), dKz = Z^T dz as
    split-bf16 exact products; vs fp64 numpy.  The error is that of an fp32 summation (no product rounding at all)."""
    from clvae_amd import ops
    rng = np.random.default_rng(K + nz)
    N, nx = 352, 88
    ldx = 92 if nz else 88
    XZ = np.zeros((K, ldx), np.float32)
    XZ[:, :nx] = (rng.random((K, nx)) < 0.0443) if exact else rng.standard_normal((K, nx))
    if nz:
        XZ[:, nx:nx + min(nz, ldx - nx)] = rng.standard_normal((K, min(nz, ldx - nx)))
    Zsrc = XZ[:, nx:] if nz <= ldx - nx else rng.standard_normal((K, nz)).astype(np.float32)
    hs = np.tanh(rng.standard_normal((K, nh))).astype(np.float32)
    dz = (rng.standard_normal((K, N)) * np.exp(rng.standard_normal((K, 1)) * 2)).astype(np.float32)   # wide dynamic range
    Hs = np.zeros_like(hs)
    Hs[1:] = hs[:-1]
    Hs[::Tn] = 0
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    tXZ, ths, tdz = d(XZ), d(hs), d(dz)
    tZ = tXZ[:, nx:] if nz <= ldx - nx else d(Zsrc)
    ldz = ldx if nz <= ldx - nx else nz
    gx = torch.full((nx, N), 7.0, device=dev)
    gu = torch.full((nh, N), 7.0, device=dev)
    gz = torch.full((max(nz, 1), N), 7.0, device=dev)
    ws = ops.Workspace(dev)
    rq = ops.ReduceQueue(dev) if defer else None
    assert ops.lstm_wgrad_supported(N, nx, nh, nz, exact)
    assert not ops.lstm_wgrad_supported(N, nx, 100, 0, True)              # h rows beyond the producers' slots
    assert not ops.lstm_wgrad_supported(N, nx, 64, 32, False)             # nz > 8 needs the wide kernel, i.e. exact frames
    ops.lstm_wgrad(K, N, tXZ, ldx, nx, exact, ths, nh, nh, Tn, tZ if nz else None, ldz, nz, tdz, gx, gu,
                   gz if nz else None, ws, defer=rq, split_scale=2 if K % 3 == 0 else 1)      # some cases on the fine grid
    if defer:
        rq.flush()
    torch.cuda.synchronize()
    f8 = lambda a: a.astype(np.float64)
    for got, A in ((gx, XZ[:, :nx]), (gu, Hs)) + (((gz, Zsrc[:, :nz]),) if nz else ()):
        ref = f8(A).T @ f8(dz)
        mag = np.abs(f8(A)).T @ np.abs(f8(dz)) + 1e-30
        err = np.abs(got.cpu().numpy() - ref) / mag
        # fp32 accumulation over K terms of piece products that are exact; an h / z row's product leaves out the piece pairs
        # below 2^-25 of it (csrc/wgrad_bf16.hip, WB_PRODUCTS): less than the rounding of one fp32 multiply
        assert err.max() < 2e-6, err.max()


@pytest.mark.parametrize("K", [64, 4096])
def test_six_of_nine_piece_pairs_stay_within_fp32_rounding_on_cancelling_inputs(dev, K):
    """The h / z / non-byte x rows of clv_lstm_wgrad multiply 6 of the 9 bf16 piece pairs of an fp32 x fp32 product
    (csrc/wgrad_bf16.hip: what is dropped is <= 2^-25 |a.b| per product).  Inputs that CANCEL -- consecutive rows of dz are
    (v, -v (1 + 2^-12)), every value with a full 24-bit mantissa, so the exact sums are ~2^-12 of the sum of magnitudes --
    against float64: the error stays below an explicit multiple of the sum of |a.b| (truncation 2^-25 + fp32 accumulation),
    i.e. it is an fp32 GEMM's error, not a bf16 one (one bf16 piece alone would be off by 2^-9)."""
    from clvae_amd import ops
    rng = np.random.default_rng(K)
    N, nx, nh, Tn = 352, 88, 88, K                       # one window: H' is hs shifted by one row
    X = rng.standard_normal((K, nx)).astype(np.float32)  # not byte-valued: three pieces, 6 pairs
    hs = np.repeat(np.tanh(rng.standard_normal((K // 2, nh))), 2, axis=0).astype(np.float32)
    X[1::2] = X[0::2]                                    # pairs of equal rows on the A side ...
    v = rng.standard_normal((K // 2, N))
    dz = np.empty((K, N), np.float32)
    dz[0::2], dz[1::2] = v, -v * (1 + 2.0 ** -12)        # ... times (v, -v(1 + 2^-12)) on the B side
    Hs = np.zeros_like(hs)
    Hs[1:] = hs[:-1]
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    gx, gu = torch.zeros(nx, N, device=dev), torch.zeros(nh, N, device=dev)
    ops.lstm_wgrad(K, N, d(X), nx, nx, False, d(hs), nh, nh, Tn, None, 0, 0, d(dz), gx, gu, None, ops.Workspace(dev))
    torch.cuda.synchronize()
    f8 = lambda a: a.astype(np.float64)
    bound = 2.0 ** -25 + np.sqrt(K) * 2.0 ** -24         # dropped pairs + a random-walk fp32 accumulation over K terms
    for name, got, A in (("dKx", gx, X), ("dU", gu, Hs)):
        ref, mag = f8(A).T @ f8(dz), np.abs(f8(A)).T @ np.abs(f8(dz))
        err = (np.abs(got.cpu().numpy() - ref) / mag).max()
        print("K=%d %s: max |err| / sum|a.b| = %.2e (bound %.2e; cancellation: |sum| / sum|a.b| = %.1e)"
              % (K, name, err, bound, np.abs(ref).max() / mag.max()))
        assert err <= bound, (name, err, bound)


@pytest.mark.parametrize("K,Tn,nz,exact,defer,scale", [(32768, 128, 2, True, True, 1),      # configuration 3: 16 stages per workgroup
                                                       (4096, 16, 2, True, False, 1), (1000, 8, 8, False, True, 2),
                                                       (96, 96, 0, False, False, 1)])
def test_lstm_wgrad_pair_matches_fp64(dev, K, Tn, nz, exact, defer, scale):
    """clv_lstm_wgrad_pair: the encoder's (no latent rows) and the decoder's (nz latent rows) kernel gradients of a step in
    ONE launch, against fp64 numpy with the single launch's bar; deferred and immediate reductions; bit-reproducible."""
    from clvae_amd import ops
    rng = np.random.default_rng(K + nz + 1)
    N, nx, nh = 352, 88, 88
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32), device=dev)
    probs, refs = [], []
    for z in (0, nz):                                   # "encoder", "decoder"
        ldx = nx + (4 if 0 < z <= 4 else z)
        XZ = np.zeros((K, ldx), np.float32)
        XZ[:, :nx] = (rng.random((K, nx)) < 0.0443) if exact else rng.standard_normal((K, nx))
        XZ[:, nx:nx + z] = rng.standard_normal((K, z))
        hs = np.tanh(rng.standard_normal((K, nh))).astype(np.float32)
        dz = (rng.standard_normal((K, N)) * np.exp(rng.standard_normal((K, 1)) * 2)).astype(np.float32)
        Hs = np.zeros_like(hs)
        Hs[1:] = hs[:-1]
        Hs[::Tn] = 0
        tXZ = d(XZ)
        g = [torch.full((nx, N), 7.0, device=dev), torch.full((nh, N), 7.0, device=dev), torch.full((max(z, 1), N), 7.0, device=dev)]
        probs.append((K, N, tXZ, ldx, nx, exact, d(hs), nh, nh, Tn, tXZ[:, nx:] if z else None, ldx, z, d(dz), g[0], g[1],
                      g[2] if z else None))
        refs.append([(g[0], XZ[:, :nx], dz), (g[1], Hs, dz)] + ([(g[2], XZ[:, nx:nx + z], dz)] if z else []))
    if not ops.lstm_wgrad_pair_supported(*probs):
        assert nz > 8 or (nz and not exact and nh + nz > 96)       # different forms of the kernel: two launches instead
        pytest.skip("the two problems take different kernels")
    outs = []
    for rep in range(2):
        rq = ops.ReduceQueue(dev) if defer else None
        ops.lstm_wgrad_pair(probs[0], probs[1], (ops.Workspace(dev), ops.Workspace(dev)), defer=rq, split_scale=scale)
        if rq is not None:
            assert rq.n == 2
            rq.flush()
        torch.cuda.synchronize()
        outs.append([t[0].clone() for r in refs for t in r])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    f8 = lambda a: a.astype(np.float64)
    for r in refs:
        for got, A, dz in r:
            ref = f8(A).T @ f8(dz)
            mag = np.abs(f8(A)).T @ np.abs(f8(dz)) + 1e-30
            err = np.abs(got.cpu().numpy() - ref) / mag
            assert err.max() < 2e-6, err.max()


@pytest.mark.parametrize("K,N,R,with_jobs", [(256, 352, 5, True), (1, 352, 5, False), (300, 100, 15, False),
                                              (1000, 64, 1, True), (37, 353, 9, False)])
def test_reduce_launch_skinny_riders(dev, K, N, R, with_jobs):
    """clv_splitk_reduce_multi: two few-row products A[:, :R]^T B (+ the column sums of B) from rider blocks of the
    reduction launch -- the label rows and biases of both LSTM input-kernel gradients (cl_vrnn/model.py:194,223) --
    with and without pending reductions and loss means in the same launch, ragged N / K and one chunk and several."""
    from clvae_amd import ops
    rng = np.random.default_rng(K + N)
    lda = R + 2
    A = rng.standard_normal((K, lda)).astype(np.float32)
    Bs = [rng.standard_normal((K, N)).astype(np.float32) for _ in range(2)]
    Cs = [torch.full((R + 1, N + 3), -7.0, dtype=torch.float32, device=dev) for _ in range(2)]
    bias = [torch.full((N,), -7.0, dtype=torch.float32, device=dev) for _ in range(2)]
    Ad, Bd = T(A, dev), [T(b, dev) for b in Bs]
    rq, ws = ops.ReduceQueue(dev), ops.Workspace(dev)
    out = torch.zeros(8, dtype=torch.float32, device=dev)
    means, ref_c = None, None
    if with_jobs:
        M2, N2, K2 = 40, 48, 5000
        a2, b2 = rng.standard_normal((K2, M2)).astype(np.float32), rng.standard_normal((K2, N2)).astype(np.float32)
        c2 = torch.zeros(M2, N2, dtype=torch.float32, device=dev)
        ops.gemm(T(a2, dev), T(b2, dev), c2, M2, N2, K2, ta=True, split_k=8, ws=ws, defer=rq)
        ref_c = a2.astype(np.float64).T @ b2
        means = [(Bd[0], K * N, 1)]
    skinny = [dict(A=Ad, lda=lda, rows=R, B=Bd[i], ldb=N, N=N, K=K, C=Cs[i], ldc=N + 3, bias_row=bias[i] if i == 0 else None)
              for i in range(2)]
    rq.flush(means=means, out=out, skinny=skinny)
    torch.cuda.synchronize()
    for i in range(2):
        ref = A[:, :R].astype(np.float64).T @ Bs[i]
        got = Cs[i].cpu().numpy()
        np.testing.assert_allclose(got[:R, :N], ref, rtol=1e-5, atol=1e-5 * np.sqrt(K))
        assert (got[R:, :] == -7.0).all() and (got[:, N:] == -7.0).all()
    np.testing.assert_allclose(bias[0].cpu().numpy(), Bs[0].astype(np.float64).sum(0), rtol=1e-5, atol=1e-5 * np.sqrt(K))
    assert (bias[1].cpu().numpy() == -7.0).all()
    if with_jobs:
        np.testing.assert_allclose(c2.cpu().numpy(), ref_c, rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(out[0].item(), Bs[0].astype(np.float64).mean(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("pads", [(3, 1), (4, 8)])
@pytest.mark.parametrize("R,L,defer", [(1, 2, False), (37, 9, False), (128, 16, True), (1000, 17, True), (4096, 32, True),
                                       (128 * 300 + 5, 32, True), (300, 1, False), (5, 4, False), (77, 20, True), (130, 12, False)])
def test_latent_head_matches_numpy(dev, R, L, defer, pads):
    """clv_latent_head_fwd / _bwd (cl_vrnn/model.py:200-216, 243 and their gradients) against fp64 numpy: zargs, the
    sample, the rows' KL; dzargs, dh_enc, dWz, dbz -- ragged row counts, one block per workgroup and the persistent
    multi-block case, both padded widths (latent_dim <= 16 / <= 32), immediate and deferred reduction, Z and dZ strided.
    pads: the strides of Z and dZ beyond latent_dim -- (4, 8) with latent_dim a multiple of 4 is what configuration 5 runs
    (every row 16-byte aligned), (3, 1) never is."""
    from clvae_amd import ops
    rng = np.random.default_rng(R + L)
    H = 88
    hs = np.tanh(rng.standard_normal((R, H))).astype(np.float32)
    Wz = (rng.standard_normal((H, 2 * L)) * 0.2).astype(np.float32)
    bz = (rng.standard_normal(2 * L) * 0.3).astype(np.float32)
    eps = rng.standard_normal((R, L)).astype(np.float32)
    ldz, lddz = L + pads[0], L + pads[1]
    z = lambda *sh: torch.full(sh, -7.0, dtype=torch.float32, device=dev)
    zargs, Z, rowkl = z(R, 2 * L), z(R, ldz), z(R)
    hs_d, Wz_d, eps_d = T(hs, dev), T(Wz, dev), T(eps, dev)
    ops.latent_head_fwd(R, H, L, hs_d, Wz_d, T(bz, dev), eps_d, zargs, Z, ldz, rowkl)
    torch.cuda.synchronize()
    za = hs.astype(np.float64) @ Wz + bz
    m, lv = za[:, :L], za[:, L:]
    sd = np.exp(0.5 * lv)
    np.testing.assert_allclose(N(zargs), za, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(N(Z)[:, :L], m + sd * eps, rtol=1e-5, atol=3e-5)
    assert (N(Z)[:, L:] == -7.0).all()
    np.testing.assert_allclose(N(rowkl), -0.5 * (1 + lv - m * m - sd * sd).sum(1), rtol=2e-5, atol=2e-4)
    # backward from the device's own zargs (what the step does)
    dZ = np.full((R, lddz), 55.0, dtype=np.float32)
    dZ[:, :L] = rng.standard_normal((R, L)) / R
    kl = 0.37 / R
    dzargs, dhs, dWz, dbz = z(R, 2 * L), z(R, H), z(H, 2 * L), z(2 * L)
    ws = ops.Workspace(dev)
    rq = ops.ReduceQueue(dev) if defer else None
    ops.latent_head_bwd(R, H, L, hs_d, Wz_d, zargs, eps_d, T(dZ, dev), lddz, kl, dhs, dWz, dbz, ws, dzargs=dzargs, defer=rq)
    if rq is not None:
        rq.flush()
    torch.cuda.synchronize()
    zd = N(zargs).astype(np.float64)
    m, lv = zd[:, :L], zd[:, L:]
    sd = np.exp(0.5 * lv)
    d = dZ[:, :L].astype(np.float64)
    dz_ref = np.concatenate([d + kl * m, d * eps * 0.5 * sd - 0.5 * kl * (1 - sd * sd)], axis=1)
    sc = 1.0 / R
    np.testing.assert_allclose(N(dzargs), dz_ref, rtol=2e-5, atol=2e-6 * sc * 10)
    np.testing.assert_allclose(N(dhs), dz_ref @ Wz.astype(np.float64).T, rtol=1e-4, atol=2e-5 * sc * 10)
    np.testing.assert_allclose(N(dWz), hs.astype(np.float64).T @ dz_ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(N(dbz), dz_ref.sum(0), rtol=1e-4, atol=2e-5)
    # the same launch without a dzargs buffer
    dhs2, dWz2, dbz2 = z(R, H), z(H, 2 * L), z(2 * L)
    ops.latent_head_bwd(R, H, L, hs_d, Wz_d, zargs, eps_d, T(dZ, dev), lddz, kl, dhs2, dWz2, dbz2, ws)
    torch.cuda.synchronize()
    assert torch.equal(dhs2, dhs) and torch.equal(dWz2, dWz) and torch.equal(dbz2, dbz)


@pytest.mark.parametrize("R,L,first", [(4096, 32, 0), (128 * 9 + 5, 32, 4096 * 32), (77, 20, 40), (130, 12, 12 * 7), (37, 9, 0),
                                       (300, 1, 5), (64, 8, 6), (1000, 17, 3), (5, 4, (1 << 33) + 8)])
def test_latent_head_draws_its_own_noise(dev, R, L, first):
    """clv_latent_head_fwd(noise=...): eps is drawn inside the launch -- element (r, l) = the Philox normal at index
    first + r * L + l (oracle/philox.py), BIT FOR BIT what clv_philox_normal writes (quad-shared counters when L and first are multiples of 4, one
    call per element otherwise; an index beyond 2^32) -- written for the backward pass, and zargs / Z / KL are those of the
    same launch fed with that eps."""
    from clvae_amd import ops
    rng = np.random.default_rng(R * L)
    H, seed, stream, step = 88, 0x1234567890AB, 7, 3
    hs = np.tanh(rng.standard_normal((R, H))).astype(np.float32)
    Wz = (rng.standard_normal((H, 2 * L)) * 0.2).astype(np.float32)
    bz = (rng.standard_normal(2 * L) * 0.3).astype(np.float32)
    z = lambda *sh: torch.full(sh, -7.0, dtype=torch.float32, device=dev)
    hs_d, Wz_d, bz_d = T(hs, dev), T(Wz, dev), T(bz, dev)
    eps_d, zargs, Z, rowkl = z(R + 1, L), z(R, 2 * L), z(R, L + 4), z(R)
    it = torch.tensor([2], dtype=torch.int32, device=dev)         # the device step counter is added to `step`
    nz = ops.noise_draw(seed, stream, first, step - 2, it)
    ops.latent_head_fwd(R, H, L, hs_d, Wz_d, bz_d, eps_d, zargs, Z, L + 4, rowkl, noise=nz)
    torch.cuda.synchronize()
    want = OP.normal(R * L, seed, step=step, stream_id=stream, first_index=first).reshape(R, L)
    got = eps_d.cpu().numpy()
    assert (got[R] == -7.0).all()
    np.testing.assert_allclose(got[:R], want, atol=2e-5)           # (the oracle's libm differs from the device's in the last bits)
    ref = torch.empty(R * L, dtype=torch.float32, device=dev)      # ... and what the stand-alone launch writes
    ops.philox_normal(ref, R * L, seed, step, stream, first)
    assert torch.equal(ref.view(R, L), eps_d[:R])
    zargs2, Z2, rowkl2 = z(R, 2 * L), z(R, L + 4), z(R)
    ops.latent_head_fwd(R, H, L, hs_d, Wz_d, bz_d, eps_d, zargs2, Z2, L + 4, rowkl2)
    torch.cuda.synchronize()
    assert torch.equal(zargs, zargs2) and torch.equal(Z, Z2) and torch.equal(rowkl, rowkl2)


def _mx_case(rng, B, Tn, nx, nz, density, gate):
    """inputs of one clv_lstm_mx_fwd / _bwd call and the oracle's forward / backward for them"""
    H = 88
    U = O.orthogonal(rng, (H, 4 * H), np.float64) * 1.5
    Kx = rng.standard_normal((max(nx, 1), 4 * H)) * 0.7
    Kz = rng.standard_normal((max(nz, 1), 4 * H)) * 0.4
    ldx, ldz = nx + 4, nz + 3
    Xb = np.zeros((B * Tn, ldx))
    if nx:
        Xb[:, :nx] = rng.random((B * Tn, nx)) < density
        Xb[1 % (B * Tn), :nx] = 1.0                       # a frame with every note on (the list loop's long path)
        Xb[:, nx:] = 7.0                                   # padding columns must not be read as inputs
    Zb = rng.standard_normal((B * Tn, ldz))
    rb = rng.standard_normal((B, 4 * H)) * 0.3
    xs = rb[:, None, :] + np.zeros((B, Tn, 4 * H))
    if nx:
        xs = xs + (f32(Xb[:, :nx]) @ f32(Kx[:nx])).reshape(B, Tn, 4 * H)
    if nz:
        xs = xs + (f32(Zb[:, :nz]) @ f32(Kz[:nz])).reshape(B, Tn, 4 * H)
    act = 'hard_sigmoid' if gate == 0 else 'sigmoid'
    hs_ref, cache = O.lstm_forward(f32(xs), np.eye(4 * H), f32(U), np.zeros(4 * H), gate_act=act)
    return dict(H=H, U=U, Kx=Kx, Kz=Kz, ldx=ldx, ldz=ldz, Xb=Xb, Zb=Zb, rb=rb, hs_ref=hs_ref, cache=cache, act=act)


MX_CASES = [(4, 9, 88, 0, 0.05, 0), (7, 5, 88, 32, 0.05, 0), (1024, 3, 88, 32, 0.0443, 0), (1030, 2, 88, 5, 0.2, 1),
            (5, 1, 88, 17, 0.05, 0), (8, 6, 0, 32, 0.0, 0), (6, 4, 96, 2, 0.3, 0), (3, 12, 60, 0, 0.05, 1)]


@pytest.mark.parametrize("B,Tn,nx,nz,density,gate", MX_CASES)
def test_lstm_mx_fwd_bwd_match_the_oracle(dev, B, Tn, nx, nz, density, gate):
    """clv_lstm_mx_fwd / _bwd (csrc/lstm_mx.hip): the recurrent product as split-bf16 exact products on the matrix cores,
    the frame rows of the input kernel gathered inside the kernel, z_t . Kz as one more k-step; odd and even T, batches
    that are not multiples of four, frames from empty to full, no frames at all, the dZ tiles of the backward pass."""
    from clvae_amd import ops
    rng = np.random.default_rng(B * 7 + Tn + nx + nz)
    c = _mx_case(rng, B, Tn, nx, nz, density, gate)
    H = c['H']
    hs = torch.full((B * Tn, H), 9.0, device=dev)
    coef = torch.full((B * Tn, 4 * H), 9.0, device=dev)
    aux = torch.full((B * Tn, 2 * H), 9.0, device=dev)
    Ud = T(c['U'], dev)
    Xd, Zd, Kxd, Kzd = T(c['Xb'], dev), T(c['Zb'], dev), T(c['Kx'], dev), T(c['Kz'], dev)
    ops.lstm_mx_fwd(B, Tn, Xd if nx else None, c['ldx'], nx, Kxd if nx else None, Zd if nz else None, c['ldz'], nz,
                    Kzd if nz else None, T(c['rb'], dev), Ud, hs, coef, aux, gate_act=gate)
    torch.cuda.synchronize()
    np.testing.assert_allclose(N(hs).reshape(B, Tn, H), c['hs_ref'], atol=5e-6)
    # the coefficients, from the oracle's pre-activations and cell states
    Zr = c['cache']['Z'].reshape(B, Tn, 4, H)
    Cs = c['cache']['C']
    if c['act'] == 'hard_sigmoid':
        fa, da = O.hard_sigmoid, O.hard_sigmoid_grad
    else:
        fa = O.sigmoid
        da = lambda z: O.sigmoid(z) * (1 - O.sigmoid(z))
    i, f, g, o = fa(Zr[:, :, 0]), fa(Zr[:, :, 1]), np.tanh(Zr[:, :, 2]), fa(Zr[:, :, 3])
    cprev = np.concatenate([np.zeros((B, 1, H)), Cs[:, :-1]], 1)
    tc = np.tanh(Cs)
    want = np.stack([g * da(Zr[:, :, 0]), cprev * da(Zr[:, :, 1]), i * (1 - g * g), tc * da(Zr[:, :, 3])], 2)
    got = N(coef).reshape(B, Tn, H, 4).transpose(0, 1, 3, 2)       # the record is unit-major: [unit][ki, kf, kg, ko]
    # a hard-sigmoid pre-activation within fp32 noise of a kink may fall on its other side
    bad = np.abs(got - want) > 2e-5 * (1 + np.abs(want))
    assert bad.mean() < 2e-4, bad.mean()
    ga = N(aux).reshape(B, Tn, H, 2).transpose(0, 1, 3, 2)         # ... and [unit][kcarry, kc]
    np.testing.assert_allclose(ga[:, :, 0], f, atol=5e-6)
    np.testing.assert_allclose(ga[:, :, 1], o * (1 - tc * tc), atol=1e-5)

    dHs = rng.standard_normal((B, Tn, H))
    _, _, _, _, dZ_ref = O.lstm_backward(f32(dHs), c['cache'], np.eye(4 * H), f32(c['U']))
    dzsum = torch.full((B, 4 * H), 9.0, device=dev)
    dZ = torch.full((B * Tn, nz + 2), 7.0, device=dev)
    ops.lstm_mx_bwd(B, Tn, Ud, T(dHs, dev), aux, coef, dzsum, Kz=Kzd if nz else None, nz=nz, dZ=dZ if nz else None,
                    lddz=nz + 2)
    torch.cuda.synchronize()
    dz = N(coef).reshape(B, Tn, 4 * H)
    bad = np.abs(dz - dZ_ref) > 3e-5 * (1 + np.abs(dZ_ref))
    assert bad.mean() < 2e-4, bad.mean()
    np.testing.assert_allclose(N(dzsum), dz.sum(1), atol=2e-4)
    if nz:
        ref = f32(dz.reshape(B * Tn, 4 * H)) @ f32(c['Kz'][:nz]).T
        np.testing.assert_allclose(N(dZ)[:, :nz], ref, atol=3e-5 * max(1.0, np.abs(ref).max()))
        assert float(dZ[:, nz:].min()) == 7.0 and float(dZ[:, nz:].max()) == 7.0
