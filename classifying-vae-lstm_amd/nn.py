"""torch.nn face of the HIP kernels: the three building blocks the reference takes from Keras
(`Dense`, `LSTM`, and the `Lambda` sampling layers; cl_vrnn/model.py:174-234, cl_vae/model.py:141-188) as
`torch.nn.Module`s whose forward AND backward are calls into libclvae_hip.so.

torch provides the parameters, the autograd graph and the stream; no arithmetic runs in torch.  The training engines
(engine.py) do not go through these modules -- they fuse across layers -- but every kernel used here is one of theirs, so
this is also the smallest way to drive a single kernel from Python:

    dense = ClvDense(88, 88, activation='relu').cuda()
    lstm = ClvLSTMSeq(98, 88).cuda()                       # Keras gate order i, f, c, o; hard_sigmoid gates
    w = ClvLogisticNormal()(w_mean, w_log_var, eps)        # softmax([mean + exp(lv/2) eps, 0])
    z = ClvGaussianSample()(z_mean, z_log_var, eps)        # mean + exp(lv/2) eps

Weight layouts are Keras' ([in, out]; LSTM kernels [in, 4H] / [H, 4H] in gate blocks i, f, c, o), so state_dicts map
one-to-one onto the .h5 tensors.  Inputs must be contiguous float32 CUDA tensors.
"""
import math

import torch
from torch import nn
from torch.autograd import Function

from . import _lib, ops

_WS = {}


def _ws(device):
    """one scratch buffer per device for the split-K products"""
    key = str(device)
    if key not in _WS:
        _WS[key] = ops.Workspace(device, 8 << 20)
    return _WS[key]


def _check(*tensors):
    for t in tensors:
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected contiguous float32 CUDA tensors")


_ACTS = {None: _lib.ACT_NONE, 'linear': _lib.ACT_NONE, 'relu': _lib.ACT_RELU, 'sigmoid': _lib.ACT_SIGMOID}


class _DenseFn(Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, act):
        _check(x, kernel, bias)
        M, K = x.shape
        N = kernel.shape[1]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ops.gemm(x, kernel, y, M, N, K, bias=bias, act=act, ws=_ws(x.device))
        ctx.save_for_backward(x, kernel, y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, kernel, y = ctx.saved_tensors
        M, K = x.shape
        N = kernel.shape[1]
        ws = _ws(x.device)
        dpre = torch.empty_like(y)
        ops.act_grad(y.numel(), ctx.act, y, dy.contiguous(), dpre)
        dx = dk = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ops.gemm(dpre, kernel, dx, M, K, N, tb=True, ws=ws)
        if ctx.needs_input_grad[1]:
            dk = torch.empty_like(kernel)
            ops.gemm(x, dpre, dk, K, N, M, ta=True, ws=ws)
        if ctx.needs_input_grad[2]:
            db = torch.empty(N, dtype=torch.float32, device=x.device)
            ops.colsum(dpre, M, N, db, ws)
        return dx, dk, db, None


class ClvDense(nn.Module):
    """keras.layers.Dense: act(x . kernel + bias), kernel [in, out] glorot-uniform, bias zeros."""

    def __init__(self, in_features, out_features, activation=None):
        super().__init__()
        if activation not in _ACTS:
            raise ValueError("activation must be one of %s" % sorted(k for k in _ACTS if k))
        self.act = _ACTS[activation]
        lim = math.sqrt(6.0 / (in_features + out_features))
        self.kernel = nn.Parameter(torch.empty(in_features, out_features).uniform_(-lim, lim))
        self.bias = nn.Parameter(torch.zeros(out_features))

    def forward(self, x):
        lead = x.shape[:-1]
        y = _DenseFn.apply(x.reshape(-1, x.shape[-1]).contiguous(), self.kernel, self.bias, self.act)
        return y.reshape(*lead, -1)


class _LstmSeqFn(Function):
    @staticmethod
    def forward(ctx, x, kernel, rec, bias, gate_act):
        _check(x, kernel, rec, bias)
        B, T, Din = x.shape
        H = rec.shape[0]
        f = dict(dtype=torch.float32, device=x.device)
        gates = torch.empty(B * T, 4 * H, **f)
        ops.gemm(x.reshape(B * T, Din), kernel, gates, B * T, 4 * H, Din, bias=bias, ws=_ws(x.device))
        hs, cs = torch.empty(B * T, H, **f), torch.empty(B * T, H, **f)
        ops.lstm_seq_fwd(B, T, gates, None, rec, hs, cs, gates, gate_act=gate_act, H=H)     # gates: x-projection in, z out
        ctx.save_for_backward(x, kernel, rec, hs, cs, gates)
        ctx.gate_act = gate_act
        return hs.view(B, T, H)

    @staticmethod
    def backward(ctx, dhs):
        x, kernel, rec, hs, cs, gates = ctx.saved_tensors
        B, T, Din = x.shape
        H = rec.shape[0]
        ws, f = _ws(x.device), dict(dtype=torch.float32, device=x.device)
        dz = gates.clone()                       # the kernel turns the saved gate pre-activations into dz in place
        dzsum = torch.empty(B, 4 * H, **f)
        ops.lstm_seq_bwd(B, T, rec, dhs.contiguous().view(B * T, H), cs, dz, dzsum, gate_act=ctx.gate_act, H=H)
        dx = torch.empty(B * T, Din, **f)
        ops.gemm(dz, kernel, dx, B * T, Din, 4 * H, tb=True, ws=ws)
        dk, dr, db = torch.empty_like(kernel), torch.empty_like(rec), torch.empty(4 * H, **f)
        # both kernels' gradients in one pass over dz: x_t rows, and h_{t-1} rows (shift 1, zero at the window starts)
        ops.gemm_grouped_tn([dict(A=x.reshape(B * T, Din), lda=Din, M=Din, C=dk),
                             dict(A=hs, lda=H, M=H, C=dr, shift=1, zero_period=T)], 4 * H, B * T, dz, ws)
        ops.colsum(dzsum, B, 4 * H, db, ws)
        return dx.view(B, T, Din), dk, dr, db, None


class ClvLSTMSeq(nn.Module):
    """keras.layers.LSTM(H, return_sequences=True) with zero initial state: the whole window in one persistent
    sequence kernel per pass.  hidden = 88 (the reference's default, cl_vrnn/train.py:90): csrc/lstm.hip, recurrent weights in
    registers; any other width 1..1024: csrc/lstm_any.hip (same contract, recurrent weights streamed from L2)."""

    def __init__(self, input_dim, hidden=88, recurrent_activation='hard_sigmoid'):
        super().__init__()
        if not 1 <= int(hidden) <= 1024:
            raise ValueError("hidden must be in 1..1024")
        self.gate_act = {'hard_sigmoid': _lib.GATE_HARD_SIGMOID, 'sigmoid': _lib.GATE_SIGMOID}[recurrent_activation]
        lim = math.sqrt(6.0 / (input_dim + 4 * hidden))
        self.kernel = nn.Parameter(torch.empty(input_dim, 4 * hidden).uniform_(-lim, lim))
        q, _ = torch.linalg.qr(torch.randn(4 * hidden, hidden))          # orthogonal, like Keras' default
        self.recurrent_kernel = nn.Parameter(q.t().contiguous())
        b = torch.zeros(4 * hidden)
        b[hidden:2 * hidden] = 1.0                                       # unit_forget_bias
        self.bias = nn.Parameter(b)

    def forward(self, x):
        return _LstmSeqFn.apply(x.contiguous(), self.kernel, self.recurrent_kernel, self.bias, self.gate_act)


class _LogisticNormalFn(Function):
    @staticmethod
    def forward(ctx, mean, log_var, eps):
        _check(mean, log_var, eps)
        B, C1 = mean.shape
        w = torch.empty(B, C1 + 1, dtype=torch.float32, device=mean.device)
        scratch = torch.empty(B, 3, dtype=torch.float32, device=mean.device)
        ops.label_fwd(B, C1 + 1, mean, log_var, C1, eps, None, 0.0, w, scratch)
        ctx.save_for_backward(mean, log_var, eps, w)
        return w

    @staticmethod
    def backward(ctx, dw):
        mean, log_var, eps, w = ctx.saved_tensors
        B, C1 = mean.shape
        dm, dlv = torch.empty_like(mean), torch.empty_like(log_var)
        no_label = torch.zeros_like(w)           # class_weight = w_kl_weight = 0: the pure chain rule through the sample
        ops.label_bwd(B, C1 + 1, mean, log_var, C1, eps, no_label, w, dw.contiguous(), 0.0, 0.0, 0.0, 1.0, dm, dlv, C1)
        return dm, dlv, None


class ClvLogisticNormal(nn.Module):
    """The `w_sampling` Lambda (cl_vae/model.py:146-157, cl_vrnn/model.py:183-191): softmax([mean + exp(lv/2) eps, 0])."""

    def forward(self, mean, log_var, eps):
        return _LogisticNormalFn.apply(mean.contiguous(), log_var.contiguous(), eps.contiguous())


class _GaussFn(Function):
    @staticmethod
    def forward(ctx, mean, log_var, eps):
        _check(mean, log_var, eps)
        R, L = mean.shape
        zargs = torch.cat([mean, log_var], dim=1).contiguous()           # the kernels take [mean | log_var] rows
        z = torch.empty(R, L, dtype=torch.float32, device=mean.device)
        ops.gauss_fwd(R, L, zargs, eps, z, L, None)
        ctx.save_for_backward(zargs, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        zargs, eps = ctx.saved_tensors
        R, L = eps.shape
        dzargs = torch.empty_like(zargs)
        ops.gauss_bwd(R, L, zargs, eps, dz.contiguous(), L, 0.0, dzargs)
        return dzargs[:, :L], dzargs[:, L:], None


class ClvGaussianSample(nn.Module):
    """The `sampling` Lambda (cl_vae/model.py:170-174, cl_vrnn/model.py:212-216): mean + exp(log_var/2) eps."""

    def forward(self, mean, log_var, eps):
        lead = mean.shape[:-1]
        f = lambda t: t.reshape(-1, t.shape[-1]).contiguous()
        return _GaussFn.apply(f(mean), f(log_var), f(eps)).reshape(*lead, -1)
