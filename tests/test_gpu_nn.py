"""-m gpu: the torch.nn face (clvae_amd/nn.py) -- forward and backward of each module against plain PyTorch fp64
autograd of the same formula on the CPU (A.1-A.3 of SURVEY.md: Keras layouts, gate order i,f,c,o, hard_sigmoid)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    _lib.require_gpu()
    return torch.device("cuda:0")


def _cmp(got, ref, tol, what):
    g, r = got.detach().cpu().double(), ref.detach().double()
    err = (g - r).abs().max().item() / (r.abs().max().item() + 1e-12)
    assert err < tol, "%s: rel err %.3e" % (what, err)


@pytest.mark.parametrize("act", [None, 'relu', 'sigmoid'])
@pytest.mark.parametrize("M,K,N", [(100, 88, 88), (37, 98, 18), (3000, 88, 352)])
def test_dense(dev, act, M, K, N):
    from clvae_amd.nn import ClvDense
    torch.manual_seed(M + N)
    layer = ClvDense(K, N, activation=act).to(dev)
    with torch.no_grad():
        layer.bias.copy_(0.1 * torch.randn(N))
    x = torch.randn(M, K, device=dev, requires_grad=True)
    g = torch.randn(M, N, device=dev)
    y = layer(x)
    y.backward(g)
    xr = x.detach().cpu().double().requires_grad_(True)
    kr, br = layer.kernel.detach().cpu().double().requires_grad_(True), layer.bias.detach().cpu().double().requires_grad_(True)
    pre = xr @ kr + br
    yr = pre if act is None else (torch.relu(pre) if act == 'relu' else torch.sigmoid(pre))
    yr.backward(g.cpu().double())
    _cmp(y, yr, 2e-6, "y")
    _cmp(x.grad, xr.grad, 2e-5, "dx")
    _cmp(layer.kernel.grad, kr.grad, 2e-5, "dkernel")
    _cmp(layer.bias.grad, br.grad, 2e-5, "dbias")
    # leading axes are kept (TimeDistributed use)
    assert layer(torch.randn(4, 5, K, device=dev)).shape == (4, 5, N)


def _lstm_ref(x, k, r, b, gate):
    B, T, _ = x.shape
    H = r.shape[0]
    h = torch.zeros(B, H, dtype=torch.float64)
    c = torch.zeros(B, H, dtype=torch.float64)
    hs = lambda z: torch.clamp(0.2 * z + 0.5, 0.0, 1.0)
    g_ = hs if gate == 'hard_sigmoid' else torch.sigmoid
    out = []
    for t in range(T):
        z = x[:, t] @ k + b + h @ r
        i, f, g, o = g_(z[:, :H]), g_(z[:, H:2 * H]), torch.tanh(z[:, 2 * H:3 * H]), g_(z[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out.append(h)
    return torch.stack(out, 1)


@pytest.mark.parametrize("gate", ['hard_sigmoid', 'sigmoid'])
@pytest.mark.parametrize("B,T,Din,H", [(3, 5, 98, 88), (16, 32, 100, 88), (2, 1, 120, 88), (4, 6, 98, 40), (3, 4, 100, 128)])
def test_lstm_seq(dev, gate, B, T, Din, H):
    from clvae_amd.nn import ClvLSTMSeq
    torch.manual_seed(B * 100 + T)
    layer = ClvLSTMSeq(Din, H, recurrent_activation=gate).to(dev)
    x = (0.5 * torch.randn(B, T, Din, device=dev)).requires_grad_(True)
    g = torch.randn(B, T, H, device=dev)
    y = layer(x)
    y.backward(g)
    ps = [p.detach().cpu().double().requires_grad_(True) for p in (layer.kernel, layer.recurrent_kernel, layer.bias)]
    xr = x.detach().cpu().double().requires_grad_(True)
    yr = _lstm_ref(xr, *ps, gate)
    yr.backward(g.cpu().double())
    _cmp(y, yr, 5e-6, "hs")
    _cmp(x.grad, xr.grad, 1e-4, "dx")
    for name, p, pr in zip(("kernel", "recurrent_kernel", "bias"), (layer.kernel, layer.recurrent_kernel, layer.bias), ps):
        _cmp(p.grad, pr.grad, 1e-4, "d" + name)


def test_sampling_layers(dev):
    from clvae_amd.nn import ClvGaussianSample, ClvLogisticNormal
    torch.manual_seed(1)
    B, C1, L = 50, 9, 6
    mean = torch.randn(B, C1, device=dev, requires_grad=True)
    lv = (0.3 * torch.randn(B, C1, device=dev)).requires_grad_(True)
    eps = torch.randn(B, C1, device=dev)
    gw = torch.randn(B, C1 + 1, device=dev)
    w = ClvLogisticNormal()(mean, lv, eps)
    w.backward(gw)
    mr, lr_ = mean.detach().cpu().double().requires_grad_(True), lv.detach().cpu().double().requires_grad_(True)
    pre = torch.cat([mr + torch.exp(lr_ / 2) * eps.cpu().double(), torch.zeros(B, 1, dtype=torch.float64)], 1)
    wr = torch.softmax(pre, dim=1)
    wr.backward(gw.cpu().double())
    _cmp(w, wr, 2e-6, "w")
    _cmp(mean.grad, mr.grad, 2e-5, "dmean")
    _cmp(lv.grad, lr_.grad, 2e-5, "dlogvar")

    zm = torch.randn(4, 7, L, device=dev, requires_grad=True)
    zl = (0.3 * torch.randn(4, 7, L, device=dev)).requires_grad_(True)
    ez, gz = torch.randn(4, 7, L, device=dev), torch.randn(4, 7, L, device=dev)
    z = ClvGaussianSample()(zm, zl, ez)
    z.backward(gz)
    a, b = zm.detach().cpu().double().requires_grad_(True), zl.detach().cpu().double().requires_grad_(True)
    zr = a + torch.exp(b / 2) * ez.cpu().double()
    zr.backward(gz.cpu().double())
    _cmp(z, zr, 2e-6, "z")
    _cmp(zm.grad, a.grad, 2e-6, "dzmean")
    _cmp(zl.grad, b.grad, 2e-5, "dzlogvar")


def test_modules_compose_into_a_trainable_graph(dev):
    """Dense -> LSTM -> Dense -> Gaussian sample under torch.optim: the loss goes down (autograd plumbing end to end)."""
    from clvae_amd.nn import ClvDense, ClvGaussianSample, ClvLSTMSeq
    torch.manual_seed(0)
    B, T = 8, 12
    net = torch.nn.ModuleDict(dict(inp=ClvDense(88, 98, 'relu'), lstm=ClvLSTMSeq(98), mean=ClvDense(88, 4),
                                   lv=ClvDense(88, 4), out=ClvDense(4, 88))).to(dev)
    x = (torch.rand(B, T, 88, device=dev) < 0.05).float()
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    losses = []
    for step in range(40):
        h = net['lstm'](net['inp'](x))
        z = ClvGaussianSample()(net['mean'](h), net['lv'](h), torch.randn(B, T, 4, device=dev))
        loss = torch.nn.functional.binary_cross_entropy_with_logits(net['out'](z), x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.6 * losses[0]
