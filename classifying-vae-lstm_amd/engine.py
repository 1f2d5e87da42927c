"""Device engines: one training / inference step of cl_vae and cl_vrnn as a chain of
HIP kernels over flat, HBM-resident parameter / gradient buffers.

The math follows cl_vae/model.py:130-219 and cl_vrnn/model.py:164-264 of the
reference; torch only provides device memory and the
stream.  Every buffer is allocated once in __init__, so a step enqueues kernels
only and can be captured into a hipGraph (ops.Graph).

Parameter layout: tensors in the order of ``param_shapes`` (the Keras
layer/weight order), each starting on a 16-byte boundary of one flat fp32
buffer; gradients, Adam m/v use the same layout; the weight-norm column state
(s, m_g, v_g) is a second flat layout over the last axis of every matrix.
"""
import contextlib
import ctypes as C
import os
import warnings

import numpy as np
import torch

from . import _lib, ops
from .engine_dropout import VrnnDropout
from .engine_generate import VaeGenerate, VrnnGenerate
from .ops import ACT_MASKPOS, ACT_NONE, ACT_RELU, ACT_SIGMOID
from .params import FlatParams, fuse_heads  # noqa: F401  (engine.FlatParams stays importable)


def vae_param_shapes(cfg):
    """Keras layer order of cl_vae.get_model (cl_vae/model.py:141-188).  intermediate_dim == 0 (:165-167,188): no `h`
    and no `decoder_h` layer, the latent heads read [x, w] and the output layer reads [w, (history,) z]."""
    D, H, L, Hc, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['Hc'], cfg['C']
    dec_in = Cn + (D if cfg['use_x_prev'] else 0) + L
    head = [('h_w/kernel', (D, Hc)), ('h_w/bias', (Hc,)),
            ('w_mean/kernel', (Hc, Cn - 1)), ('w_mean/bias', (Cn - 1,)),
            ('w_log_var/kernel', (Hc, Cn - 1)), ('w_log_var/bias', (Cn - 1,))]
    if H > 0:
        return head + [('h/kernel', (D + Cn, H)), ('h/bias', (H,)),
                       ('z_mean/kernel', (H, L)), ('z_mean/bias', (L,)),
                       ('z_log_var/kernel', (H, L)), ('z_log_var/bias', (L,)),
                       ('decoder_h/kernel', (dec_in, H)), ('decoder_h/bias', (H,)),
                       ('x_decoded_mean/kernel', (H, D)), ('x_decoded_mean/bias', (D,))]
    return head + [('z_mean/kernel', (D + Cn, L)), ('z_mean/bias', (L,)),
                   ('z_log_var/kernel', (D + Cn, L)), ('z_log_var/bias', (L,)),
                   ('x_decoded_mean/kernel', (dec_in, D)), ('x_decoded_mean/bias', (D,))]


def vrnn_param_shapes(cfg):
    """Keras layer order of cl_vrnn.get_model (cl_vrnn/model.py:174-234)."""
    D, H, L, T, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
    dec_in = (D if cfg['use_x_prev'] else 0) + L + Cn
    return [('hW/kernel', (T * D, D)), ('hW/bias', (D,)),
            ('Wargs/kernel', (D, 2 * (Cn - 1))), ('Wargs/bias', (2 * (Cn - 1),)),
            ('encoder_h/kernel', (D + Cn, 4 * H)), ('encoder_h/recurrent_kernel', (H, 4 * H)),
            ('encoder_h/bias', (4 * H,)),
            ('Z_mean/kernel', (H, L)), ('Z_mean/bias', (L,)),
            ('Z_log_var/kernel', (H, L)), ('Z_log_var/bias', (L,)),
            ('decoder_h/kernel', (dec_in, 4 * H)), ('decoder_h/recurrent_kernel', (H, 4 * H)),
            ('decoder_h/bias', (4 * H,)),
            ('X_decoded_mean/kernel', (H, D)), ('X_decoded_mean/bias', (D,))]


# Which kernels a VrnnEngine's chains take.  Every entry is an engine cfg key (tests, A/B sessions: cfg[key] = True / False), an
# environment variable of the same meaning for runs that build their engines elsewhere (bench.py, the CLIs), and the default --
# the product configuration, what every number in DESIGN.md was measured with.  Each is narrowed by what the shapes support
# (the *_supported() predicates of include/clvae.h); tests/test_gpu_switches.py walks every one of them through a two-step
# oracle check in its non-default position.
SWITCHES = {
    # key:            (env variable,           default, what the non-default position selects)
    'fuse_pair':      (None,                   True,  "the separate sequence kernels (csrc/lstm.hip) instead of encoder + latent head + decoder as one launch"),
    'sparse_inputs':  (None,                   True,  "dense GEMM input projections instead of row gathers of the notes that are on"),
    'keep_logits':    (None,                   True,  "the fused output head does not store its logits (what a replayed training step does: 11.5 / 92 MB per step)"),
    'fuse_head':      ('CLV_FUSE_HEAD',        True,  "output head as GEMM + NLL + three backward GEMMs instead of one launch"),
    'bf16_wgrad':     ('CLV_BF16_WGRAD',       True,  "the LSTM kernel gradients on the f32 MFMA GEMM instead of split-bf16 products"),
    'wgrad_pair':     ('CLV_WGRAD_PAIR',       True,  "one kernel-gradient launch per LSTM instead of both in one"),
    'dense_hw_grad':  ('CLV_DENSE_HW_GRAD',    True,  "the hW kernel gradient by walking the notes instead of dense on the bf16 matrix cores"),
    'dense_hw_fwd':   ('CLV_DENSE_HW_FWD',     True,  "the hW forward product by walking the notes at every batch size"),
    'label_in_pair':  ('CLV_LABEL_IN_PAIR',    True,  "the label path's backward as a launch of its own instead of the pair backward kernel's epilogue"),
    'fuse_latent':    ('CLV_FUSE_LATENT',      True,  "the latent head as GEMM + pointwise launches instead of csrc/latent_head.hip"),
    'lstm_mx':        ('CLV_USE_MX',           True,  "large batches on the generic chain instead of csrc/lstm_mx.hip"),
    'frames_u8':      ('CLV_FRAMES_U8',        True,  "the large-batch step widens its byte batch to float (rounds 4-5) instead of reading bytes"),
    'front_fused':    ('CLV_FRONT_FUSED',      True,  "the frame projections as a launch of their own behind the label launch instead of workgroups of it (csrc/label_head.hip: vrnn_front_kernel)"),
    'fine_grid':      ('CLV_FINE_GRID',        False, "twice the workgroups, half the rows each, for the LSTM kernel gradients (what tune_dp_schedule may pick next to an all-reduce)"),
    'fuse_notes':     ('CLV_FUSE_NOTES',       False, "input projections gathered inside the pair forward from note lists (+48 us on MI355X: profiles/r03_notes_fusion_ab.txt)"),
}


def switch(cfg, key):
    """cfg[key] if given, else the switch's environment variable ('0' / '1'), else its default"""
    env, default, _ = SWITCHES[key]
    if key in cfg:
        return bool(cfg[key])
    if env is not None and env in os.environ:
        return os.environ[env] != '0'
    return default


def _f(device, *shape):
    return torch.empty(*shape, dtype=torch.float32, device=device)


class _EngineBase:
    def __init__(self, cfg, batch_size, shapes, device, head_pairs=(), grads_pre=0):
        _lib.require_gpu()
        self.cfg = dict(cfg)
        self.B = int(batch_size)
        self.device = torch.device(device)
        phys, aliases = fuse_heads(shapes, head_pairs)
        self.P = FlatParams(shapes, self.device, phys, aliases, pre=grads_pre)
        self.ws = ops.Workspace(self.device, 8 << 20)
        self.ws2 = ops.Workspace(self.device, 8 << 20)        # scratch of the side stream
        self.rq = ops.ReduceQueue(self.device)                # split-K reductions of the weight gradients, one launch per pass
        self.side = torch.cuda.Stream(device=self.device) if cfg.get('two_streams', False) else None
        self.scal = torch.zeros(8, dtype=torch.float32, device=self.device)   # vae, kl_z, kl_w, w_rec, acc
        # loss weights may be annealed per epoch (utils/model_utils.py:19-50)
        self.kl_weight = float(cfg.get('kl_weight', 1.0))
        self.w_kl_weight = float(cfg.get('w_kl_weight', 1.0))
        self.class_weight = float(cfg.get('class_weight', 1.0))

    def losses(self):
        """Host copy of the last step's loss terms (one small D2H)."""
        s = self.scal.detach().cpu().numpy().astype(np.float64)
        out = dict(vae=s[0], kl_z=s[1], kl_w=s[2], w_rec=s[3], acc=s[4])
        out['total'] = (out['vae'] + self.w_kl_weight * out['kl_w'] + self.class_weight * out['w_rec']
                        + self.kl_weight * out['kl_z'])
        out['elbo'] = -(out['vae'] + out['kl_z'] + out['kl_w'] + out['w_rec'])
        return out

    def _mean_into(self, n, x, stride, slot):
        ops.sum_strided(n, x, stride, 1.0 / n, self.scal[slot:])

    # -- two-stream DAG: MFMA GEMMs on the side stream overlap the VALU-bound LSTM kernels ---------
    def _side(self):
        """Context: enqueue on the side stream, ordered after everything enqueued so far on the main one."""
        if self.side is None:
            return contextlib.nullcontext()
        self.side.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.side)

    def _side_more(self):
        """Context: continue on the side stream without a new dependency on the main stream."""
        return contextlib.nullcontext() if self.side is None else torch.cuda.stream(self.side)

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


# --------------------------------------------------------------------------- #
class VaeEngine(VaeGenerate, _EngineBase):
    """cl_vae: Dense encoder/decoder VAE with a logistic-normal label (cl_vae/model.py:130-224).

    Concatenations ([x,w], [w,xp,z]) are never materialised: a Dense over a concatenation is
    the sum of GEMMs over row blocks of its kernel (beta = 1 accumulation)."""

    def __init__(self, cfg, batch_size, device='cuda:0'):
        super().__init__(cfg, batch_size, vae_param_shapes(cfg), device,
                         head_pairs=[('w_mean', 'w_log_var', 'wargs'), ('z_mean', 'z_log_var', 'zargs')])
        B, D, H, L, Hc, Cn = self.B, cfg['D'], cfg['H'], cfg['L'], cfg['Hc'], cfg['C']
        d = self.device
        self.xoff = D if cfg['use_x_prev'] else 0      # decoder_h kernel rows: [w | xp | z]
        L_ = _lib.lib()
        self.fused = H > 0 and bool(cfg.get('fused_step', True)) and bool(L_.clv_vae_fused_supported(D, H, Hc, Cn, L))
        self.stage_spec = None       # set by TrainStep for one fused step: see _fused_step
        names = ['h_w', 'wargs', 'h', 'zargs', 'decoder_h', 'x_decoded_mean']
        self._offs = (C.c_int64 * 12)(*[self.P.offsets.get('%s/%s' % (n, w), 0) for n in names for w in ('kernel', 'bias')])
        ws_bytes = L_.clv_vae_fused_workspace_bytes(B, D, H, Hc, Cn, L, int(cfg['use_x_prev'])) if self.fused else 0
        self._fused_ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.device) if self.fused else None
        self.h_w = _f(d, B, Hc)
        self.wargs = _f(d, B, 2 * (Cn - 1))          # [w_mean | w_log_var]
        self.w = _f(d, B, Cn)
        self.rowloss = _f(d, B, 3)
        self.h = _f(d, B, max(H, 1))
        self.zargs = _f(d, B, 2 * L)
        self.z = _f(d, B, L)
        self.h_dec = _f(d, B, max(H, 1))
        self.logits = _f(d, B, D)
        self.dlogits = _f(d, B, D)
        self.rownll = _f(d, B)
        self.rowkl = _f(d, B)
        self.d_hdec = _f(d, B, max(H, 1))
        self.dz = _f(d, B, L)
        self.dzargs = _f(d, B, 2 * L)
        self.d_h = _f(d, B, max(H, 1))
        self.dw = _f(d, B, Cn)
        self.dwargs = _f(d, B, 2 * (Cn - 1))
        self.d_hw = _f(d, B, Hc)

    # -- forward pieces (also used by the predict() sub-models) ----------------
    def encode_w(self, x, B=None):
        """h_w, w_mean, w_log_var (:141-143) -> self.wargs"""
        cfg, P = self.cfg, self.P
        B = self.B if B is None else B
        D, Hc, C1 = cfg['D'], cfg['Hc'], cfg['C'] - 1
        g = ops.gemm
        g(x, P.p('h_w/kernel'), self.h_w, B, Hc, D, bias=P.p('h_w/bias'), act=ACT_RELU, ws=self.ws)
        g(self.h_w, P.p('wargs/kernel'), self.wargs, B, 2 * C1, Hc, bias=P.p('wargs/bias'), ws=self.ws)

    def encode_z(self, x, w, B=None):
        """h = relu([x,w].K + b); z_mean, z_log_var (:160-164) -> self.zargs"""
        cfg, P = self.cfg, self.P
        B = self.B if B is None else B
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g = ops.gemm
        if H == 0:      # :165-167: the latent heads on [x, w] directly
            g(x, P.p('zargs/kernel'), self.zargs, B, 2 * L, D, ws=self.ws)
            g(w, P.rows(P.params, 'zargs/kernel', D), self.zargs, B, 2 * L, Cn, beta=1.0, bias=P.p('zargs/bias'), ws=self.ws)
            return
        g(x, P.p('h/kernel'), self.h, B, H, D, ws=self.ws)
        g(w, P.rows(P.params, 'h/kernel', D), self.h, B, H, Cn, beta=1.0, bias=P.p('h/bias'), act=ACT_RELU,
          ws=self.ws)
        g(self.h, P.p('zargs/kernel'), self.zargs, B, 2 * L, H, bias=P.p('zargs/bias'), ws=self.ws)

    def decode(self, w, z, xp, B=None, act=ACT_NONE):
        """logits (or x_hat with act=sigmoid) = Dense(relu([w,xp,z].K + b)) (:177-188) -> self.logits"""
        cfg, P = self.cfg, self.P
        B = self.B if B is None else B
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g = ops.gemm
        if H == 0:      # :188: the output layer on [w, (history,) z] directly
            K = 'x_decoded_mean/kernel'
            g(w, P.p(K), self.logits, B, D, Cn, ws=self.ws)
            if cfg['use_x_prev']:
                g(xp, P.rows(P.params, K, Cn), self.logits, B, D, D, beta=1.0, ws=self.ws)
            g(z, P.rows(P.params, K, Cn + self.xoff), self.logits, B, D, L, beta=1.0, bias=P.p('x_decoded_mean/bias'),
              act=act, ws=self.ws)
            return
        g(w, P.p('decoder_h/kernel'), self.h_dec, B, H, Cn, ws=self.ws)
        if cfg['use_x_prev']:
            g(xp, P.rows(P.params, 'decoder_h/kernel', Cn), self.h_dec, B, H, D, beta=1.0, ws=self.ws)
        g(z, P.rows(P.params, 'decoder_h/kernel', Cn + self.xoff), self.h_dec, B, H, L, beta=1.0,
          bias=P.p('decoder_h/bias'), act=ACT_RELU, ws=self.ws)
        g(self.h_dec, P.p('x_decoded_mean/kernel'), self.logits, B, D, H, bias=P.p('x_decoded_mean/bias'), act=act,
          ws=self.ws)

    def x_hat(self):
        """sigmoid(logits) of the last forward -> self.dlogits (the Keras output `x_decoded_mean`)."""
        cfg, P, B = self.cfg, self.P, self.B
        if cfg['H'] == 0:       # no hidden decoder activations to restart from: run the output layer again with the sigmoid
            self.decode(self.w, self.z, self._last_xp, act=ACT_SIGMOID)
            return self.logits
        ops.gemm(self.h_dec, P.p('x_decoded_mean/kernel'), self.dlogits, B, cfg['D'], cfg['H'],
                 bias=P.p('x_decoded_mean/bias'), act=ACT_SIGMOID, ws=self.ws)
        return self.dlogits

    def forward(self, x, xp, eps_w, eps_z, w_true=None):
        cfg, B = self.cfg, self.B
        self._last_xp = xp
        L, Cn = cfg['L'], cfg['C']
        C1 = Cn - 1
        self.encode_w(x)
        ops.label_fwd(B, Cn, self.wargs, self.wargs[:, C1:], 2 * C1, eps_w, w_true, cfg['w_log_var_prior'],
                      self.w, self.rowloss)
        self.encode_z(x, self.w)
        ops.gauss_fwd(B, L, self.zargs, eps_z, self.z, L, self.rowkl)
        self.decode(self.w, self.z, xp)

    def _fused_step(self, x, xp, w_true, eps_w, eps_z, need_grads, target=None, noise=None, bump=False):
        """The whole step as ONE kernel (csrc/vae_fused.hip) + one launch that sums the gradient slabs, takes the loss
        means and (bump) advances the step counter."""
        cfg, P, B = self.cfg, self.P, self.B
        p_ = ops._ptr
        opts = _lib.VaeStepOpts()
        opts.loss_means = p_(self.scal)
        opts.bf16 = int(bool(cfg.get('bf16', False)))
        if noise is not None:
            (opts.noise_seed, opts.stream_w, opts.stream_z, opts.first_w, opts.first_z, opts.step), step_dev = noise[:6], noise[6]
            opts.draw, opts.step_dev = 1, p_(step_dev)
        if bump and need_grads:
            opts.bump_iterations = p_(P.iterations)
        tail = (p_(eps_w), p_(eps_z), p_(P.params), self._offs, P.n, float(cfg['w_log_var_prior']), self.class_weight,
                self.kl_weight, self.w_kl_weight, int(need_grads), p_(P.grads), p_(self._fused_ws), self._fused_ws.numel(),
                p_(self.logits), p_(self.w), p_(self.wargs), p_(self.zargs), p_(self.rownll), p_(self.rowkl),
                p_(self.rowloss), C.byref(opts), ops._stream())
        stage, self.stage_spec = self.stage_spec, None
        # stage: TrainStep handed over the mini-batch assembly (ops.label_stage): the kernel reads its rows' byte frames itself
        staged = stage is not None and target is None
        _lib.check(_lib.lib().clv_vae_fused_step(
            B, cfg['D'], cfg['H'], cfg['Hc'], cfg['C'], cfg['L'], int(cfg['use_x_prev']), p_(x), p_(xp), p_(target), p_(w_true),
            C.byref(stage) if staged else None, *tail), "clv_vae_fused_step")

    def can_stage_in_label(self):
        """Can the step's own launch assemble its mini-batch (ops.label_stage)?  The fused kernel reads its rows itself."""
        return bool(self.fused)

    def folds_step(self, w_true):
        """True when loss_and_grads(noise=..., bump=True) draws the noise and advances `iterations` inside the step's
        own launches (the fused kernel): the caller then skips its own draw and passes advanced=True to adam_step."""
        return bool(self.fused and w_true is not None)

    def loss_and_grads(self, x, xp, w_true, eps_w, eps_z, need_grads=True, target=None, noise=None, bump=False):
        """One forward + 4 losses (+ gradients of the weighted total into P.grads).  target: what the decoder output is
        scored against (default x; the next frame under --predict_next).
        noise = (seed, stream_w, stream_z, first_w, first_z, step, step_dev): draw eps_w / eps_z (into the buffers
        passed) instead of reading them; bump: advance P.iterations once the gradients are in (only honoured together
        with folds_step())."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, Hc, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['Hc'], cfg['C']
        C1 = Cn - 1
        inv = 1.0 / B
        if self.fused and w_true is not None:
            return self._fused_step(x, xp, w_true, eps_w, eps_z, need_grads, target, noise, bump)
        if noise is not None:
            ops.philox_normal2(eps_w, B * C1, noise[1], noise[3], eps_z, B * L, noise[2], noise[4], noise[0], noise[5],
                               step_dev=noise[6])
        self.forward(x, xp, eps_w, eps_z, w_true)
        ops.bernoulli_nll(B, D, self.logits, x if target is None else target, D, inv, self.rownll,
                          self.dlogits if need_grads else None)
        ops.loss_sums([(self.rownll, B, 1), (self.rowkl, B, 1), (self.rowloss, B, 3), (self.rowloss[:, 1:], B, 3),
                       (self.rowloss[:, 2:], B, 3)], self.scal)
        if not need_grads:
            return
        g, ws, xo = ops.gemm, self.ws, self.xoff
        if H == 0:
            return self._grads_without_hidden_layers(x, xp, w_true, eps_w, eps_z, inv)
        # decoder
        g(self.h_dec, self.dlogits, P.g('x_decoded_mean/kernel'), H, D, B, ta=True, ws=ws)
        ops.colsum(self.dlogits, B, D, P.g('x_decoded_mean/bias'), ws)
        g(self.dlogits, P.p('x_decoded_mean/kernel'), self.d_hdec, B, H, D, tb=True, act=ACT_MASKPOS, aux=self.h_dec,
          ws=ws)
        g(self.w, self.d_hdec, P.g('decoder_h/kernel'), Cn, H, B, ta=True, ws=ws)
        if cfg['use_x_prev']:
            g(xp, self.d_hdec, P.rows(P.grads, 'decoder_h/kernel', Cn), D, H, B, ta=True, ws=ws)
        g(self.z, self.d_hdec, P.rows(P.grads, 'decoder_h/kernel', Cn + xo), L, H, B, ta=True, ws=ws)
        ops.colsum(self.d_hdec, B, H, P.g('decoder_h/bias'), ws)
        g(self.d_hdec, P.p('decoder_h/kernel'), self.dw, B, Cn, H, tb=True, ws=ws)
        g(self.d_hdec, P.rows(P.params, 'decoder_h/kernel', Cn + xo), self.dz, B, L, H, tb=True, ws=ws)
        # latent heads
        ops.gauss_bwd(B, L, self.zargs, eps_z, self.dz, L, self.kl_weight * inv, self.dzargs)
        g(self.h, self.dzargs, P.g('zargs/kernel'), H, 2 * L, B, ta=True, ws=ws)
        ops.colsum(self.dzargs, B, 2 * L, P.g('zargs/bias'), ws)
        g(self.dzargs, P.p('zargs/kernel'), self.d_h, B, H, 2 * L, tb=True, act=ACT_MASKPOS, aux=self.h, ws=ws)
        g(x, self.d_h, P.g('h/kernel'), D, H, B, ta=True, ws=ws)
        g(self.w, self.d_h, P.rows(P.grads, 'h/kernel', D), Cn, H, B, ta=True, ws=ws)
        ops.colsum(self.d_h, B, H, P.g('h/bias'), ws)
        g(self.d_h, P.rows(P.params, 'h/kernel', D), self.dw, B, Cn, H, tb=True, beta=1.0, ws=ws)
        self._label_head_grads(x, w_true, eps_w, inv)


    def _label_head_grads(self, x, w_true, eps_w, inv):
        """label head backward from self.dw (shared by both variants)"""
        cfg, P, B = self.cfg, self.P, self.B
        D, Hc, Cn = cfg['D'], cfg['Hc'], cfg['C']
        C1, g, ws = Cn - 1, ops.gemm, self.ws
        ops.label_bwd(B, Cn, self.wargs, self.wargs[:, C1:], 2 * C1, eps_w, w_true, self.w, self.dw,
                      cfg['w_log_var_prior'], self.class_weight, self.w_kl_weight, inv,
                      self.dwargs, self.dwargs[:, C1:], 2 * C1)
        g(self.h_w, self.dwargs, P.g('wargs/kernel'), Hc, 2 * C1, B, ta=True, ws=ws)
        ops.colsum(self.dwargs, B, 2 * C1, P.g('wargs/bias'), ws)
        g(self.dwargs, P.p('wargs/kernel'), self.d_hw, B, Hc, 2 * C1, tb=True, act=ACT_MASKPOS, aux=self.h_w, ws=ws)
        g(x, self.d_hw, P.g('h_w/kernel'), D, Hc, B, ta=True, ws=ws)
        ops.colsum(self.d_hw, B, Hc, P.g('h_w/bias'), ws)

    def _grads_without_hidden_layers(self, x, xp, w_true, eps_w, eps_z, inv):
        """Backward of the intermediate_dim == 0 graph (cl_vae/model.py:165-167,188): logits = [w, xp, z].K_o + b_o and
        zargs = [x, w].K_z + b_z, no relu layers in between."""
        cfg, P, B = self.cfg, self.P, self.B
        D, L, Cn = cfg['D'], cfg['L'], cfg['C']
        g, ws, xo, K = ops.gemm, self.ws, self.xoff, 'x_decoded_mean/kernel'
        g(self.w, self.dlogits, P.g(K), Cn, D, B, ta=True, ws=ws)
        if cfg['use_x_prev']:
            g(xp, self.dlogits, P.rows(P.grads, K, Cn), D, D, B, ta=True, ws=ws)
        g(self.z, self.dlogits, P.rows(P.grads, K, Cn + xo), L, D, B, ta=True, ws=ws)
        ops.colsum(self.dlogits, B, D, P.g('x_decoded_mean/bias'), ws)
        g(self.dlogits, P.p(K), self.dw, B, Cn, D, tb=True, ws=ws)
        g(self.dlogits, P.rows(P.params, K, Cn + xo), self.dz, B, L, D, tb=True, ws=ws)
        ops.gauss_bwd(B, L, self.zargs, eps_z, self.dz, L, self.kl_weight * inv, self.dzargs)
        g(x, self.dzargs, P.g('zargs/kernel'), D, 2 * L, B, ta=True, ws=ws)
        g(self.w, self.dzargs, P.rows(P.grads, 'zargs/kernel', D), Cn, 2 * L, B, ta=True, ws=ws)
        ops.colsum(self.dzargs, B, 2 * L, P.g('zargs/bias'), ws)
        g(self.dzargs, P.rows(P.params, 'zargs/kernel', D), self.dw, B, Cn, 2 * L, tb=True, beta=1.0, ws=ws)
        self._label_head_grads(x, w_true, eps_w, inv)


# --------------------------------------------------------------------------- #
class VrnnEngine(VrnnDropout, VrnnGenerate, _EngineBase):
    """cl_vrnn: classifying VAE + two LSTMs (cl_vrnn/model.py:164-267)."""

    def __init__(self, cfg, batch_size, device='cuda:0'):
        super().__init__(cfg, batch_size, vrnn_param_shapes(cfg), device,
                         head_pairs=[('Z_mean', 'Z_log_var', 'Zargs')], grads_pre=cfg['D'])
        B, D, H, L, T, Cn = self.B, cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
        # H = --intermediate_dim (cl_vrnn/train.py:90).  The fused and the matrix-core paths below are laid out for the
        # reference's default of 88 units and say so through their *_supported() predicates; any other width takes the
        # generic chain: GEMM (or row-gather) input projections + csrc/lstm_any.hip + the GEMM heads.
        d = self.device
        BT = B * T
        if H != 88:       # once per width (the warnings filter's default): a user of --intermediate_dim should know what runs
            warnings.warn("cl_vrnn with %d LSTM units runs on the generic kernels (csrc/lstm_any.hip, GEMM heads): several times "
                          "slower per step than the default 88 units, which the fused kernels are laid out for" % H, stacklevel=2)
        self.gate_act = _lib.GATE_HARD_SIGMOID if cfg.get('gate_act', 'hard_sigmoid') == 'hard_sigmoid' \
            else _lib.GATE_SIGMOID
        self.off = D if cfg['use_x_prev'] else 0     # decoder kernel rows: [Xp | Z | W]
        # LSTM(dropout=p) (cl_vrnn/model.py:164,198,227; never set by the reference's scripts): input dropout in TRAINING passes
        # (loss_and_grads(need_grads=True)), one mask per gate and sample, constant over the time steps (Keras 2.0.0,
        # implementation 0).  Taken by the generic chain only: per-gate masked projections as GEMMs (_forward_dropout).
        self.dropout = float(cfg.get('dropout', 0.0) or 0.0)
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError("dropout must be in [0, 1)")
        self._train_pass = False
        # which kernels the chains take: SWITCHES (cfg key -> what it selects), each narrowed by what the shapes support
        sw = lambda key: switch(cfg, key)
        self.fuse_pair = sw('fuse_pair') and ops.lstm_pair_supported(L, H) and not self.dropout
        self.fine_grid = sw('fine_grid')
        self.pair_pack = _f(d, ops.lstm_pair_pack_floats()) if self.fuse_pair else None
        self.sparse_inputs = sw('sparse_inputs') and ops.sparse_proj_supported(D, 4 * H)
        self.fuse_head = sw('fuse_head') and ops.out_head_train_supported(H, D)
        self._head_done = False
        self.keep_logits = sw('keep_logits')
        self.bf16_wgrad = sw('bf16_wgrad')
        # every staged frame value is exactly a bf16 number (TrainStep sets it when the data set is kept as uint8): the frame
        # rows of the kernel-gradient products then use one bf16 piece instead of three
        self.frames_exact_bf16 = bool(cfg.get('frames_exact_bf16', False))
        self.wgrad_pair = sw('wgrad_pair')
        self.dense_hw_grad = sw('dense_hw_grad')
        # (from 512 batch rows on: at 256 rows the note-walking gather is as fast, 20.5 against 12.5 + 11.7 us)
        self.dense_hw_fwd = sw('dense_hw_fwd') and B >= int(os.environ.get('CLV_DENSE_HW_FWD_ROWS', '512'))
        self.ws_hw = None
        self.stage_spec = None       # set by TrainStep for one forward pass: see _forward_pair
        self._f8 = None              # the pass's uint8 frames (forward(frames8=...))
        self.ws_b = None
        self.label_in_pair = sw('label_in_pair') and self.fuse_pair
        self.fuse_latent = sw('fuse_latent') and ops.latent_head_supported(H, L)
        self.use_mx = sw('lstm_mx') and not self.fuse_pair and not self.dropout and self.sparse_inputs \
            and ops.lstm_mx_supported(B, D, L, H)
        self.frames_u8 = sw('frames_u8')
        self.front_fused = sw('front_fused')
        # notes_valid: the note lists describe the frames now in X / XZ (TrainStep sets it per staged batch)
        self.fuse_notes = sw('fuse_notes') and self.fuse_pair and self.sparse_inputs and D == ops.NOTE_NONE
        self.notes_valid = False
        if self.fuse_notes:
            self.notes_enc = torch.full((BT, ops.NOTE_ROW), ops.NOTE_NONE, dtype=torch.uint8, device=d)
            self.notes_dec = torch.full((BT, ops.NOTE_ROW), ops.NOTE_NONE, dtype=torch.uint8, device=d)
        self.hW = _f(d, B, D)
        self.wargs = _f(d, B, 2 * (Cn - 1))
        self.W = _f(d, B, Cn)
        self.rowloss = _f(d, B, 3)
        self.wk_enc = _f(d, B, 4 * H)
        self.wk_dec = _f(d, B, 4 * H)
        self.gates_enc = _f(d, BT, 4 * H)           # xproj in, (z_i,z_f,g,z_o) after fwd, dz after bwd
        self.gates_dec = _f(d, BT, 4 * H)
        # cs_*: the cell states of the separate sequence kernels; the pair kernels keep (kcarry, kc) per unit and step
        # there instead (two floats: see csrc/lstm_pair.hip), so the buffers hold 2H floats per frame
        aux2 = self.fuse_pair or self.use_mx
        self.hs_enc, self.cs_enc = _f(d, BT, H), _f(d, BT, 2 * H if aux2 else H)
        self.hs_dec, self.cs_dec = _f(d, BT, H), _f(d, BT, 2 * H if aux2 else H)
        self.zargs = _f(d, BT, 2 * L)
        # decoder input [Xp | Z] as ONE matrix (row stride padded to a multiple of 4 floats): the history
        # frames are staged into its first D columns, gauss_fwd writes Z next to them, so the decoder's
        # input projection and its kernel gradient are single GEMMs over K = D + L
        self.xz_ld = (self.off + L + 3) // 4 * 4
        self.XZ = torch.zeros(BT, self.xz_ld, dtype=torch.float32, device=d)
        self.Z = self.XZ[:, self.off:self.off + L]
        self.rowkl = _f(d, BT)
        self.klterm = _f(d, BT, L)                  # fused pair kernel: L * KL_l per latent (mean over all = per-frame KL)
        self.logits = _f(d, BT, D)
        self.dlogits = _f(d, BT, D)
        self.rownll = _f(d, BT)
        self.dhs = _f(d, BT, H)                     # dL/dh of the decoder, then of the encoder
        self.dzsum_enc, self.dzsum_dec = _f(d, B, 4 * H), _f(d, B, 4 * H)
        self.dZ = _f(d, BT, L)
        self.dzargs = _f(d, BT, 2 * L)
        self.dW = _f(d, B, Cn)
        self.dwargs = _f(d, B, 2 * (Cn - 1))
        self.dhW = _f(d, B, D)
        # sum_j hW/kernel[j,c] * its gradient[j,c] (FlatParams.adam_step): in front of the gradient buffer, so that the
        # data-parallel step averages it with the hW kernel's bucket
        self.gdot = self.P.grads_pre[:D] if (self.P.grads_pre.numel() >= D and self.P.offsets['hW/kernel'] == 0) else _f(d, D)
        self.gdot_fresh = False
        if self.dropout:
            in_e, in_d = D + Cn, self.off + L + Cn
            self.u_enc, self.u_dec = _f(d, B, 4, in_e), _f(d, B, 4, in_d)          # the masks uniforms [row][gate][input] (clv_dropout_rows)
            self._masks_given = False      # u_enc / u_dec hold nothing until a pass draws them (noise=...) or set_dropout_uniforms()
            # a gate's masked per-step inputs.  xm_d's rows are padded to xz_ld floats and the projection GEMM reads whole float4s of
            # a row: the padding columns must be ZERO (they meet the kernel's label rows), which clv_dropout_rows never writes --
            # torch.empty here passed every test on a fresh device and failed behind any process that had left other bytes there
            self.xm_e = _f(d, 4, BT, D)
            self.xm_d = torch.zeros(4, BT, self.xz_ld, dtype=torch.float32, device=d)
            self.wm_e, self.wm_d = _f(d, 4, B, Cn), _f(d, 4, B, Cn)                # ... and masked label rows
            self.dxg, self.dwg = _f(d, BT, self.xz_ld), _f(d, B, Cn)               # a gate's share of dL/d[Xp | Z], dL/dW

    def folds_noise(self):
        """True when forward(noise=...) draws eps_W / eps_Z inside the label kernel and the pair kernels / the latent head of the
        large-batch path (no Philox launch)."""
        return bool((self.fuse_pair or (self.use_mx and self.fuse_latent)) and self.sparse_inputs and self.cfg['D'] % 2 == 0)

    def frames_u8_route(self):
        """How a training pass can get its frames as BYTES (frames8 of forward / loss_and_grads / grads_tail: a uint8 batch copied out
        of the frame store, never widened to float), or None.  Every consumer of frames must be on its byte-reading kernel:
        csrc/lstm_mx.hip (note lists) or sparse_proj.hip (row gathers), wgrad_bf16.hip (x rows), out_head_bf16.hip (targets),
        outer_bf16.hip (the hW layer's products).
          'gather': the large-batch path -- the staging launch copies the bytes (TrainStep._stage_bound);
          'label':  the pair path -- the label forward launch, which assembles the mini-batch anyway (ops.label_stage), leaves
                    the byte batch instead of the float one (its own note walk reads the frame store)."""
        cfg, B = self.cfg, self.B
        D, H, L, T = cfg['D'], cfg['H'], cfg['L'], cfg['T']
        if not (self.frames_u8 and self.fuse_head and self.bf16_wgrad and self.dense_hw_grad and D % 4 == 0 and self.sparse_inputs
                and ops.dense_outer_bf16_supported(B, T * D, D, T * D, D) and ops.sparse_dense_supported(D)
                and ops.lstm_wgrad_supported(4 * H, D, H, 0, 2) and (not self.off or ops.lstm_wgrad_supported(4 * H, D, H, L, 2))):
            return None
        if self.use_mx:
            return 'gather' if (self.dense_hw_fwd and ops.dense_window_fwd_bf16_supported(B, T * D, D, T * D, D)) else None
        if self.fuse_pair and self.can_stage_in_label():
            return 'label'
        return None

    def frames_u8_supported(self):
        return self.frames_u8_route() is not None

    def forward(self, X, Xp, eps_W, eps_Z, w_true=None, keep_gates=True, nll=None, target=None, noise=None, frames8=None):
        """nll = (scale, need_grads): fuse the Bernoulli NLL of the output head into its GEMM; target = the frames the
        output is scored against (default X).
        frames8 = (X8, Xp8): uint8 [B,T,D] copies of the batch's current / history frames (Xp8 None without history) -- the pass
        then reads its frames from THEM, as bytes, and X / Xp are not read at all (frames_u8_supported(); training passes
        whose target is X).
        noise = (seed, stream_w, stream_z, first_w, first_z, step, step_dev): draw eps_W / eps_Z (into the buffers passed)
        instead of reading them -- inside the label and pair kernels where they run, else with one Philox launch."""
        if frames8 is not None and (target is not None or not self.frames_exact_bf16 or not self.frames_u8_supported()):
            raise ValueError("frames8 needs the large-batch path (frames_u8_supported()), frames_exact_bf16 and the input frames as "
                             "the target")
        self._f8 = frames8
        target = X if target is None else target
        cfg, P, B = self.cfg, self.P, self.B
        self._nll_done = False
        self._noise = None
        if noise is not None:
            if self.folds_noise():
                seed, sw, sz, fw, fz, step, step_dev = noise
                self._noise = (ops.noise_draw(seed, sw, fw, step, step_dev), ops.noise_draw(seed, sz, fz, step, step_dev))
            else:
                ops.philox_normal2(eps_W, B * (cfg['C'] - 1), noise[1], noise[3], eps_Z, B * cfg['T'] * cfg['L'], noise[2],
                                   noise[4], noise[0], noise[5], step_dev=noise[6])
        D, H, L, T, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
        C1, BT, G4 = Cn - 1, B * T, 4 * H
        g, ws = ops.gemm, self.ws
        if cfg['use_x_prev'] and frames8 is None and Xp.data_ptr() != self.XZ.data_ptr():
            self.XZ.view(B, T, self.xz_ld)[:, :, :D].copy_(Xp.view(B, T, D))     # staging copy only
        if self.fuse_pair:
            return self._forward_pair(X, eps_W, eps_Z, w_true, nll, target)
        if self.use_mx:
            return self._forward_mx(X, eps_W, eps_Z, w_true, nll, target)
        if self.dropout and self._train_pass:
            if noise is not None:      # this step's masks: uniforms of the streams 2 (encoder) and 3 (decoder), by GLOBAL row
                row0 = int(noise[3]) // max(cfg['C'] - 1, 1)
                for u, sid in ((self.u_enc, 2), (self.u_dec, 3)):
                    per_row = u.shape[1] * u.shape[2]
                    ops.philox_uniform(u, B * per_row, noise[0], noise[5], sid, row0 * per_row, step_dev=noise[6])
                self._masks_given = True
            elif not self._masks_given:
                raise RuntimeError("training pass with dropout=%g and neither noise=... nor set_dropout_uniforms(): the masks "
                                   "would come from uninitialised memory" % self.dropout)
            return self._forward_dropout(X, eps_W, eps_Z, w_true, nll, target)
        if self.sparse_inputs:     # add the kernel rows of the notes that are on (csrc/sparse_proj.hip)
            ops.sparse_proj(BT, D, G4, X, D, P.p('encoder_h/kernel'), self.gates_enc)
        else:
            with self._side():
                g(X, P.p('encoder_h/kernel'), self.gates_enc, BT, G4, D, ws=self.ws2)
        # label path (:174-191)
        off = self.off
        self._label_forward(X, eps_W, w_true)
        self._join()
        ops.lstm_seq_fwd(B, T, self.gates_enc, self.wk_enc, P.p('encoder_h/recurrent_kernel'), self.hs_enc,
                         self.cs_enc, self.gates_enc, gate_act=self.gate_act, H=H)
        # latent heads + sample (:200-216)
        if self.fuse_latent:       # one launch on the matrix cores (csrc/latent_head.hip)
            ops.latent_head_fwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), P.p('Zargs/bias'), eps_Z, self.zargs, self.Z,
                                self.xz_ld, self.rowkl)
        else:
            g(self.hs_enc, P.p('Zargs/kernel'), self.zargs, BT, 2 * L, H, bias=P.p('Zargs/bias'), ws=ws)
            ops.gauss_fwd(BT, L, self.zargs, eps_Z, self.Z, self.xz_ld, self.rowkl)
        # decoder LSTM on [Xp, Z, repeat(W)] (:218-228): one projection of the [Xp | Z] rows
        self._join()
        g(self.XZ, P.p('decoder_h/kernel'), self.gates_dec, BT, G4, off + L, lda=self.xz_ld, ws=ws)
        ops.lstm_seq_fwd(B, T, self.gates_dec, self.wk_dec, P.p('decoder_h/recurrent_kernel'), self.hs_dec,
                         self.cs_dec, self.gates_dec, gate_act=self.gate_act, H=H)
        # output head (:229-234), with the NLL fused into its epilogue when the caller wants the loss
        self._output_head(target, nll)

    def _forward_mx(self, X, eps_W, eps_Z, w_true, nll, target):
        """Forward for large batches: label path, encoder LSTM, latent head, decoder LSTM, output head; the LSTMs' input
        products (frame rows gathered from LDS, z_t . K_z as one more MFMA k-step) run inside csrc/lstm_mx.hip."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T = cfg['D'], cfg['H'], cfg['L'], cfg['T']
        BT, off = B * T, self.off
        f8 = self._f8
        if f8 is not None:       # the frames as bytes: the batch the staging launch copied out of the frame store
            X = f8[0]
        self._label_forward(X, eps_W, w_true)
        ops.lstm_mx_fwd(B, T, X, D, D, P.p('encoder_h/kernel'), None, 0, 0, None, self.wk_enc,
                        P.p('encoder_h/recurrent_kernel'), self.hs_enc, self.gates_enc, self.cs_enc, gate_act=self.gate_act)
        if self.fuse_latent:
            nz = getattr(self, '_noise', None)          # (set by forward(noise=...) when folds_noise(): eps_Z is drawn in the launch)
            ops.latent_head_fwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), P.p('Zargs/bias'), eps_Z, self.zargs, self.Z,
                                self.xz_ld, self.rowkl, noise=nz[1] if nz else None)
        else:
            ops.gemm(self.hs_enc, P.p('Zargs/kernel'), self.zargs, BT, 2 * L, H, bias=P.p('Zargs/bias'), ws=self.ws)
            ops.gauss_fwd(BT, L, self.zargs, eps_Z, self.Z, self.xz_ld, self.rowkl)
        hist, hist_ld = (f8[1], D) if (f8 is not None and off) else (self.XZ if off else None, self.xz_ld)
        ops.lstm_mx_fwd(B, T, hist, hist_ld, off, P.p('decoder_h/kernel') if off else None,
                        self.Z, self.xz_ld, L, P.rows(P.params, 'decoder_h/kernel', off), self.wk_dec,
                        P.p('decoder_h/recurrent_kernel'), self.hs_dec, self.gates_dec, self.cs_dec, gate_act=self.gate_act)
        self._output_head(X if f8 is not None else target, nll)

    def _dense_hw_fwd_now(self):
        cfg = self.cfg
        return self.dense_hw_fwd and self.frames_exact_bf16 and \
            ops.dense_window_fwd_bf16_supported(self.B, cfg['T'] * cfg['D'], cfg['D'], cfg['T'] * cfg['D'], cfg['D'])

    def can_stage_in_label(self):
        """Can the label forward launch assemble the step's mini-batch itself (ops.label_stage; TrainStep decides per bound
        batch source)?  The fused pair path with the note-walking label kernel: its workgroup per batch row reads the row's
        byte frames for its scan anyway."""
        return self.fuse_pair and self.sparse_inputs and self.cfg['D'] % 2 == 0 and not self.fuse_notes and \
            not (self.dense_hw_fwd and ops.dense_window_fwd_bf16_supported(self.B, self.cfg['T'] * self.cfg['D'], self.cfg['D'],
                                                                         self.cfg['T'] * self.cfg['D'], self.cfg['D']))

    def _label_forward(self, X, eps_W, w_true, pack=None, stage=None, proj=None):
        """Label path (:174-191): hW Dense layer over the flattened window, Wargs head, logistic-normal sample, label
        losses and both per-row LSTM biases (W.K_w + b).  One launch when the window is handled sparsely."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C'], self.off
        G4 = 4 * H
        tail = (P.p('Wargs/kernel'), P.p('Wargs/bias'), eps_W, w_true, cfg['w_log_var_prior'],
                P.rows(P.params, 'encoder_h/kernel', D), P.p('encoder_h/bias'),
                P.rows(P.params, 'decoder_h/kernel', off + L), P.p('decoder_h/bias'),
                self.wargs, self.W, self.rowloss, self.wk_enc, self.wk_dec)
        if self.sparse_inputs and D % 2 == 0:
            nz = getattr(self, '_noise', None)
            parts = None
            if stage is None and self._dense_hw_fwd_now():
                # byte-valued frames: the product dense on the bf16 matrix cores (split-K partial sums, summed by the label
                # launch) instead of the note-walking gather
                if self.ws_hw is None:
                    self.ws_hw = ops.Workspace(self.device)
                parts = ops.dense_window_fwd_bf16(B, T * D, D, X, T * D, P.p('hW/kernel'), D, self.ws_hw)
            ops.vrnn_label_fwd_x(B, D, Cn, G4, X, T * D, T * D, P.p('hW/kernel'), P.p('hW/bias'), self.hW, *tail,
                                 noise=nz[0] if nz else None, pack=pack, parts=parts, stage=stage, proj=proj)
        else:
            ops.gemm(X, P.p('hW/kernel'), self.hW, B, D, T * D, bias=P.p('hW/bias'), act=ACT_RELU, ws=self.ws)
            ops.vrnn_label_fwd(B, D, Cn, G4, self.hW, *tail)

    def _output_head(self, X, nll):
        """X: the frames the output is scored against"""
        cfg, P = self.cfg, self.P
        D, H, BT = cfg['D'], cfg['H'], self.B * cfg['T']
        self._head_done = False
        if nll is not None:
            scale, need_grads = nll
            if need_grads and self.fuse_head:
                # forward, loss, dL/dh_dec and the layer's weight gradients in one pass over hs_dec (csrc/out_head.hip);
                # the weight-gradient slabs are summed with the other pending reductions in grads_tail()
                ops.out_head_train(BT, H, D, self.hs_dec, P.p('X_decoded_mean/kernel'), P.p('X_decoded_mean/bias'), X, scale,
                                   self.rownll, self.dhs, P.g('X_decoded_mean/kernel'), P.g('X_decoded_mean/bias'),
                                   self.ws, logits=self.logits if self.keep_logits else None, defer=self._rq())
                self._head_done = True
            else:
                ops.gemm_bce(self.hs_dec, P.p('X_decoded_mean/kernel'), P.p('X_decoded_mean/bias'), X, scale, self.logits,
                             self.dlogits if need_grads else None, self.rownll, BT, D, H)
            self._nll_done = True
        else:
            ops.gemm(self.hs_dec, P.p('X_decoded_mean/kernel'), self.logits, BT, D, H, bias=P.p('X_decoded_mean/bias'),
                     ws=self.ws)

    def _forward_pair(self, X, eps_W, eps_Z, w_true, nll=None, target=None):
        """Forward with both LSTMs, the latent head and the z projection in one persistent kernel."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
        BT, G4, off = B * T, 4 * H, self.off
        g, ws = ops.gemm, self.ws
        # this step's recurrent kernels, K_z and the head kernel in the lane order of the pair kernels (both passes): a
        # by-product of the label kernel's launch where that runs, else a launch of its own
        pack = (L, P.p('encoder_h/recurrent_kernel'), P.p('decoder_h/recurrent_kernel'),
                P.rows(P.params, 'decoder_h/kernel', off), P.p('Zargs/kernel'), self.pair_pack)
        pack_in_label = self.sparse_inputs and D % 2 == 0
        if not pack_in_label:
            ops.lstm_pair_pack(*pack)
        # TrainStep handed over the mini-batch assembly (stage_spec: ops.label_stage): the label launch goes FIRST and fills X,
        # the history frames and the labels for everything behind it
        stage = self.stage_spec if pack_in_label else None
        self.stage_spec = None
        f8 = self._f8
        if f8 is not None and stage is None:
            raise ValueError("frames8 on the pair path: the byte batch is what the label launch's own stage leaves (ops.label_stage)")
        # the frame projections inside the label launch (csrc/label_head.hip: vrnn_front_kernel), from the byte stores the stage names
        proj = None
        if stage is not None and f8 is not None and self.front_fused and self.sparse_inputs and not self.fuse_notes and \
                ops.vrnn_label_fwd_x_proj_supported(B, D, T * D, T, G4):
            proj = (T, G4, P.p('encoder_h/kernel'), self.gates_enc, P.p('decoder_h/kernel') if off else None,
                    self.gates_dec if off else None)
        if stage is not None:
            self._label_forward(X, eps_W, w_true, pack=pack, stage=stage, proj=proj)
        notes = None
        if proj is not None:
            pass
        elif self.fuse_notes and self.notes_valid:      # the projections are gathered inside the pair kernel
            notes = (self.notes_enc, P.p('encoder_h/kernel'), self.notes_dec if off else None,
                     P.p('decoder_h/kernel') if off else None)
        elif self.sparse_inputs:     # piano-roll frames are ~4 % nonzero: add the kernel rows of the notes that are on
            cur, hist, hist_ld = (X, self.XZ, self.xz_ld) if f8 is None else (f8[0], f8[1], D)
            if off:        # both LSTMs in one launch
                ops.sparse_proj2(BT, G4, (D, cur, D, P.p('encoder_h/kernel'), self.gates_enc),
                                 (off, hist, hist_ld, P.p('decoder_h/kernel'), self.gates_dec))
            else:
                ops.sparse_proj(BT, D, G4, cur, D, P.p('encoder_h/kernel'), self.gates_enc)
        else:
            g(X, P.p('encoder_h/kernel'), self.gates_enc, BT, G4, D, ws=ws)
            if off:        # history frames only: z_t . K_z is added inside the sequence kernel
                g(self.XZ, P.p('decoder_h/kernel'), self.gates_dec, BT, G4, off, lda=self.xz_ld, ws=ws)
        if stage is None:
            self._label_forward(X, eps_W, w_true, pack=pack if pack_in_label else None)
        nz = getattr(self, '_noise', None)
        ops.lstm_pair_fwd(B, T, L, self.gates_enc, self.wk_enc, self.gates_dec, off > 0, self.wk_dec, self.pair_pack,
                          P.p('Zargs/bias'), eps_Z, self.hs_enc, self.cs_enc, self.hs_dec, self.cs_dec, self.zargs, self.Z,
                          self.xz_ld, self.klterm, gate_act=self.gate_act, noise=nz[1] if nz else None, notes=notes)
        self._output_head(f8[0] if f8 is not None else (X if target is None else target), nll)

    def xp_view(self):
        """[B,T,D] strided view of the history columns of the [Xp | Z] buffer (stage batches straight into it)."""
        return self.XZ.view(self.B, self.cfg['T'], self.xz_ld)[:, :, :self.cfg['D']]

    def x_hat(self):
        """sigmoid(logits) of the last forward -> self.dlogits (the Keras output `X_decoded_mean`)."""
        cfg, P = self.cfg, self.P
        ops.gemm(self.hs_dec, P.p('X_decoded_mean/kernel'), self.dlogits, self.B * cfg['T'], cfg['D'], cfg['H'],
                 bias=P.p('X_decoded_mean/bias'), act=ACT_SIGMOID, ws=self.ws)
        return self.dlogits

    def _wgrad_args(self, name, X_in, x_ld, x_rows, hs, dz, x8=None):
        """The argument tuple of ops.lstm_wgrad (up to dKz) for one LSTM's kernel gradients on the bf16 matrix cores
        (csrc/wgrad_bf16.hip: x rows, h rows and z rows at once), or None where that kernel does not apply.
        x8: the frame rows as a uint8 [B*T, nx] batch instead of the first nx columns of X_in (the z rows stay in X_in)."""
        cfg, P = self.cfg, self.P
        H, T, L = cfg['H'], cfg['T'], cfg['L']
        nz = L if name == 'decoder_h' else 0          # the decoder's per-step inputs are [x_{t-1} | z_t]
        nx = x_rows - nz                              # 0: a decoder without history frames
        if not (self.bf16_wgrad and nx > 0 and ops.lstm_wgrad_supported(4 * H, nx, H, nz, self.frames_exact_bf16)):
            return None
        xs, xs_ld = (X_in, x_ld) if (x8 is None or nx == 0) else (x8, nx)
        return (self.B * T, 4 * H, xs, xs_ld, nx, self.frames_exact_bf16, hs, H, H, T, X_in[:, nx:] if nz else None, x_ld, nz,
                dz, P.g(name + '/kernel'), P.g(name + '/recurrent_kernel'),
                P.rows(P.grads, name + '/kernel', nx) if nz else None)

    def _lstm_wgrads(self, name, X_in, x_ld, x_rows, hs, dz, dzsum, w_row, ws, products=True, x8=None):
        """Every weight gradient of one LSTM in two grouped launches.
        Over dz [B*T,4H] (K = B*T): kernel rows of the per-step inputs (x_t, or [x_{t-1} | z_t] for the
        decoder) and the recurrent kernel (h_{t-1}: shift 1, zero at t == 0); products=False: already formed
        (grads_tail: both LSTMs in one launch).
        Over dzsum [B,4H] (K = B): kernel rows of the repeated label W, and the bias."""
        cfg, P, B = self.cfg, self.P, self.B
        H, Cn = cfg['H'], cfg['C']
        G4 = 4 * H
        rq = self._rq()
        if products:
            args = self._wgrad_args(name, X_in, x_ld, x_rows, hs, dz, x8)
            if args is None and x8 is not None:
                raise RuntimeError("frames8: the %s kernel gradients have no byte-reading kernel at these shapes" % name)
            if args is not None:
                ops.lstm_wgrad(*args, ws, defer=rq, split_scale=2 if self.fine_grid else 1)
            else:
                self._lstm_wgrads_f32(name, X_in, x_ld, x_rows, hs, dz, ws, rq)
        if not (Cn + 1 <= 16 and B <= 4096):      # else: both LSTMs' label rows + biases in one launch (grads_tail)
            ops.gemm_grouped_tn([dict(A=self.W, lda=Cn, M=Cn, C=P.rows(P.grads, name + '/kernel', w_row)),
                                 dict(A=None, M=1, C=P.g(name + '/bias'), ones=1)], G4, B, dzsum, ws, defer=rq)

    def _lstm_wgrads_f32(self, name, X_in, x_ld, x_rows, hs, dz, ws, rq):
        """The same products on the f32 MFMA (grouped GEMM over dz): any shape."""
        cfg, P = self.cfg, self.P
        H, T = cfg['H'], cfg['T']
        BT, G4 = self.B * T, 4 * H
        ops.gemm_grouped_tn([dict(A=X_in, lda=x_ld, M=x_rows, C=P.g(name + '/kernel')),
                             dict(A=hs, lda=H, M=H, C=P.g(name + '/recurrent_kernel'), shift=1, zero_period=T)],
                            G4, BT, dz, ws, defer=rq, split_scale=2 if self.fine_grid else 1)

    def _dense_wgrad(self, name, A, lda, rows, N, K, Bm, ws, rq):
        """Kernel and bias gradient of a Dense layer in one pass over Bm = dL/d(output) [K,N].  When the bias follows the
        kernel in the flat gradient buffer the pair is ONE problem of rows+1 output rows (last row = implicit ones)."""
        P = self.P
        gk, gb = P.g(name + '/kernel'), P.g(name + '/bias')
        if gb.data_ptr() == gk.data_ptr() + 4 * rows * N and lda % 4 == 0 and A.data_ptr() % 16 == 0:
            probs = [dict(A=A, lda=lda, M=rows + 1, C=gk, ldc=N, ones=2)]
        else:
            probs = [dict(A=A, lda=lda, M=rows, C=gk), dict(A=None, M=1, C=gb, ones=1)]
        ops.gemm_grouped_tn(probs, N, K, Bm, ws, defer=rq)

    def _rq(self):
        """The deferred-reduction queue (single-stream schedule only: a pending job pins its scratch buffer)."""
        return self.rq if self.side is None else None

    def tail_range(self):
        """(offset, numel) of the hW-kernel bucket inside the flat gradient buffer: complete after
        loss_and_grads(do_tail=False), i.e. before grads_tail() has produced the rest."""
        D, T = self.cfg['D'], self.cfg['T']
        return self.P.offsets['hW/kernel'], T * D * D

    def _bptt_separate(self, eps_Z, ws):
        """Backward through both LSTMs and the latent head as separate launches (any latent_dim); leaves dz in
        gates_*, dzsum_*, dzargs."""
        cfg, P, B = self.cfg, self.P, self.B
        H, L, T = cfg['H'], cfg['L'], cfg['T']
        BT, G4, off = B * T, 4 * H, self.off
        g = ops.gemm
        if self.use_mx:      # the backward pass of csrc/lstm_mx.hip, dZ = dz_dec . Kz^T as its latent tiles
            ops.lstm_mx_bwd(B, T, P.p('decoder_h/recurrent_kernel'), self.dhs, self.cs_dec, self.gates_dec, self.dzsum_dec,
                            Kz=P.rows(P.params, 'decoder_h/kernel', off), nz=L, dZ=self.dZ, lddz=L)
        elif H == 88 and L <= 40:      # dZ = dz_dec . Kz^T by two more waves of the decoder's backward kernel
            ops.lstm_seq_bwd_z(B, T, P.p('decoder_h/recurrent_kernel'), self.dhs, self.cs_dec, self.gates_dec,
                               self.dzsum_dec, P.rows(P.params, 'decoder_h/kernel', off), L, self.dZ, L,
                               gate_act=self.gate_act)
        else:
            ops.lstm_seq_bwd(B, T, P.p('decoder_h/recurrent_kernel'), self.dhs, self.cs_dec, self.gates_dec,
                             self.dzsum_dec, gate_act=self.gate_act, H=H)
            g(self.gates_dec, P.rows(P.params, 'decoder_h/kernel', off), self.dZ, BT, L, G4, tb=True, ws=ws)
        if self.fuse_latent:       # dzargs, dh_enc and the head's own kernel / bias gradient in one launch; dzargs stays on chip
            ops.latent_head_bwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), self.zargs, eps_Z, self.dZ, L,
                                self.kl_weight / BT, self.dhs, P.g('Zargs/kernel'), P.g('Zargs/bias'), ws, defer=self._rq())
            self._head_grad_done = True
        else:
            ops.gauss_bwd(BT, L, self.zargs, eps_Z, self.dZ, L, self.kl_weight / BT, self.dzargs)
            g(self.dzargs, P.p('Zargs/kernel'), self.dhs, BT, H, 2 * L, tb=True, ws=ws)
        if self.use_mx:
            ops.lstm_mx_bwd(B, T, P.p('encoder_h/recurrent_kernel'), self.dhs, self.cs_enc, self.gates_enc, self.dzsum_enc)
        else:
            ops.lstm_seq_bwd(B, T, P.p('encoder_h/recurrent_kernel'), self.dhs, self.cs_enc, self.gates_enc,
                             self.dzsum_enc, gate_act=self.gate_act, H=H)

    def loss_and_grads(self, X, Xp, w_true, eps_W, eps_Z, need_grads=True, do_tail=True, target=None, noise=None, frames8=None):
        """target: the frames the decoder output is scored against (default X; the next frames under --predict_next).
        noise, frames8: see forward() (with frames8 and do_tail=False, hand the same frames8 to grads_tail())."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
        C1, BT, G4 = Cn - 1, B * T, 4 * H
        inv_bt, inv_b = 1.0 / BT, 1.0 / B
        g, ws, off = ops.gemm, self.ws, self.off
        self._train_pass = bool(need_grads)      # LSTM(dropout=p) applies in training passes only
        try:
            self.forward(X, Xp, eps_W, eps_Z, w_true, nll=(inv_bt, need_grads), target=target, noise=noise, frames8=frames8)
        finally:
            self._train_pass = False
        if not self._nll_done:
            ops.bernoulli_nll(BT, D, self.logits, X if target is None else target, D, inv_bt, self.rownll,
                              self.dlogits if need_grads else None)
        kl = (self.klterm, BT * L, 1) if self.fuse_pair else (self.rowkl, BT, 1)
        terms = [(self.rownll, BT, 1), kl, (self.rowloss, B, 3), (self.rowloss[:, 1:], B, 3), (self.rowloss[:, 2:], B, 3)]
        # the loss means ride in the reduce launch that ends the backward pass (grads_tail) when there is one
        self._loss_terms = terms if (need_grads and self._rq() is not None) else None
        if self._loss_terms is None:
            ops.loss_sums(terms, self.scal)
        if not need_grads:
            self._join()
            return
        # Backward, early part: everything on the critical chain dlogits -> BPTT -> label path -> hW kernel gradient.
        # The hW kernel (T*D*D floats) is 87 % of the gradient bytes: with N > 1 GPUs its all-reduce bucket starts
        # here and runs under the weight-gradient products of grads_tail().
        if not self._head_done:
            g(self.dlogits, P.p('X_decoded_mean/kernel'), self.dhs, BT, H, D, tb=True, ws=ws)
        if self.dropout:
            self._backward_dropout(X, w_true, eps_W, eps_Z)
            return
        if self.fuse_pair:
            # decoder BPTT, dZ, the latent head's backward, dh_enc and encoder BPTT: one persistent launch
            # ... and the latent head's own weight gradient (per-row slabs into the pass's pending reductions)
            # ... and (round 3) the label path's backward of the same batch row as every workgroup's epilogue: row b's
            # sum_t dz is all it needs of the BPTT, a launch of its own was 10.7 us
            label = None
            if self.label_in_pair:
                label = dict(D=D, C=Cn, Kenc_w=P.rows(P.params, 'encoder_h/kernel', D),
                             Kdec_w=P.rows(P.params, 'decoder_h/kernel', off + L), wargs=self.wargs, eps=eps_W, onehot=w_true,
                             W=self.W, hW=self.hW, Ka=P.p('Wargs/kernel'), prior=cfg['w_log_var_prior'],
                             class_weight=self.class_weight, w_kl_weight=self.w_kl_weight, inv_b=inv_b, dwargs=self.dwargs,
                             dhW=self.dhW, layer_grad=(P.g('Wargs/kernel'), P.g('Wargs/bias')))
            ops.lstm_pair_bwd(B, T, L, self.kl_weight * inv_bt, self.pair_pack, P.p('Zargs/kernel'), self.dhs,
                              self.cs_dec, self.cs_enc, self.gates_dec, self.gates_enc, self.dzsum_dec, self.dzsum_enc,
                              self.zargs, eps_Z, self.dzargs, gate_act=self.gate_act,
                              head_grad=(self.hs_enc, P.g('Zargs/kernel'), P.g('Zargs/bias')), ws=ws, defer=self._rq(),
                              label=label)
            self._head_grad_done = True
        else:
            self._bptt_separate(eps_Z, ws)
        # label head: dW from both LSTMs, label backward, dWargs and dhW in one launch
        if not (self.fuse_pair and self.label_in_pair):
            ops.vrnn_label_bwd(B, D, Cn, G4, self.dzsum_enc, self.dzsum_dec, P.rows(P.params, 'encoder_h/kernel', D),
                               P.rows(P.params, 'decoder_h/kernel', off + L), self.wargs, eps_W, w_true, self.W, self.hW,
                               P.p('Wargs/kernel'), cfg['w_log_var_prior'], self.class_weight, self.w_kl_weight, inv_b,
                               self.dwargs, self.dhW, layer_grad=(P.g('Wargs/kernel'), P.g('Wargs/bias')), ws=ws,
                               defer=self._rq())
        if self.sparse_inputs and ops.sparse_dense_supported(D):      # kernel gradient and bias gradient (column sums of dhW)
            # ... and sum_j K dK per column for the optimizer's two-launch form (the batch sum of pre-activation x gradient)
            # frames that are exactly bf16 numbers (kept as bytes): the dense product on the bf16 matrix cores, one pass
            # over X (csrc/outer_bf16.hip); else the kernel that walks the notes
            outer = ops.dense_outer_bf16 if (self.dense_hw_grad and self.frames_exact_bf16 and
                                             ops.dense_outer_bf16_supported(B, T * D, D, T * D, D)) else ops.sparse_outer
            outer(B, T * D, D, X if frames8 is None else frames8[0], T * D, self.dhW, D, P.g('hW/kernel'), colsum=P.g('hW/bias'),
                  gdot=(self.hW, D, P.p('hW/bias'), self.gdot))
            self.gdot_fresh = True
        else:
            g(X, self.dhW, P.g('hW/kernel'), T * D, D, B, ta=True, ws=ws)
            ops.colsum(self.dhW, B, D, P.g('hW/bias'), ws)
        if do_tail:
            self.grads_tail(X, frames8=frames8)

    def grads_tail(self, X, frames8=None):
        """Backward, late part: every other weight gradient (products over dz / dlogits / dzargs / dwargs that
        nothing downstream waits for), their split-K reductions in one launch.  frames8: as the pass's loss_and_grads()."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
        C1, BT, off = Cn - 1, B * T, self.off
        ws, rq = self.ws, self._rq()
        if self.dropout:           # _backward_dropout has formed every gradient
            return
        # output head: kernel and bias gradient in one pass over dlogits (bias = an implicit row of ones)
        if not self._head_done:
            self._dense_wgrad('X_decoded_mean', self.hs_dec, H, H, D, BT, self.dlogits, ws, rq)
        # both LSTMs' kernel gradients in ONE launch where they take the same form of the kernel (configuration 3: 256
        # workgroups of 16 stages instead of twice 256 of 8; half as many slabs to reduce)
        dec = ('decoder_h', self.XZ, self.xz_ld, off + L, self.hs_dec, self.gates_dec)
        enc = ('encoder_h', X, D, D, self.hs_enc, self.gates_enc)
        if frames8 is not None:      # the frame rows of both products read the byte batch (the z rows stay in [Xp | Z])
            dec, enc = dec + (frames8[1],), enc + (frames8[0],)
        paired = False
        if self.wgrad_pair:
            pd, pe = self._wgrad_args(*dec), self._wgrad_args(*enc)
            if pd is not None and pe is not None and ops.lstm_wgrad_pair_supported(pd, pe):
                if rq is None and self.ws_b is None:
                    self.ws_b = ops.Workspace(self.device)
                ops.lstm_wgrad_pair(pd, pe, (ws, self.ws_b), defer=rq, split_scale=2 if self.fine_grid else 1)
                paired = True
        self._lstm_wgrads(*dec[:6], self.dzsum_dec, off + L, ws, products=not paired, x8=dec[6] if len(dec) > 6 else None)
        if not getattr(self, '_head_grad_done', False):
            self._dense_wgrad('Zargs', self.hs_enc, H, H, 2 * L, BT, self.dzargs, ws, rq)
        self._head_grad_done = False
        self._lstm_wgrads(*enc[:6], self.dzsum_enc, D, ws, products=not paired, x8=enc[6] if len(enc) > 6 else None)
        skinny = None
        if Cn + 1 <= 16 and B <= 4096:
            # label rows + bias of both LSTMs' input-kernel gradients (K = batch rows of sum_t dz)
            if rq is not None:      # ... as rider blocks of the reduction launch
                skinny = [dict(A=self.W, lda=Cn, rows=Cn, B=dzs, ldb=4 * H, N=4 * H, K=B,
                               C=P.rows(P.grads, name + '/kernel', w_row), ldc=4 * H, bias_row=P.g(name + '/bias'))
                          for name, w_row, dzs in (('encoder_h', D, self.dzsum_enc), ('decoder_h', off + L, self.dzsum_dec))]
            else:
                wprobs = lambda name, w_row: [dict(A=self.W, lda=Cn, M=Cn, C=P.rows(P.grads, name + '/kernel', w_row)),
                                              dict(A=None, M=1, C=P.g(name + '/bias'), ones=1)]
                ops.gemm_grouped_tn_small2(wprobs('encoder_h', D), self.dzsum_enc, wprobs('decoder_h', off + L),
                                           self.dzsum_dec, 4 * H, B)
        # (the Wargs layer's gradient: per-row slabs of the label backward kernel, already among the pending reductions)
        if rq is not None:
            rq.flush(means=getattr(self, '_loss_terms', None), out=self.scal, skinny=skinny)
            self._loss_terms = None
