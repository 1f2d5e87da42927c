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
from .ops import ACT_MASKPOS, ACT_NONE, ACT_RELU, ACT_SIGMOID


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


def fuse_heads(shapes, pairs):
    """Physical layout where each (a, b, fused) pair of same-input Dense heads is ONE [in, na+nb] kernel and
    one [na+nb] bias: a single GEMM serves both heads (forward, dX and dW).  Columns are independent under
    weight normalisation and Adam, so the update is identical to two separate tensors.
    Returns (physical shapes, aliases: logical name -> (physical name, col0, ncols))."""
    d = dict(shapes)
    phys, aliases, done = [], {}, set()
    for name, shp in shapes:
        layer, w = name.split('/')
        hit = [pr for pr in pairs if layer in pr[:2]]
        if not hit:
            phys.append((name, shp))
            continue
        a, b, fused = hit[0]
        na, nb = d[a + '/kernel'][1], d[b + '/kernel'][1]
        for wn in ('kernel', 'bias'):
            aliases[a + '/' + wn] = (fused + '/' + wn, 0, na)
            aliases[b + '/' + wn] = (fused + '/' + wn, na, nb)
        if fused not in done:
            done.add(fused)
            phys.append((fused + '/kernel', (d[a + '/kernel'][0], na + nb)))
            phys.append((fused + '/bias', (na + nb,)))
    return phys, aliases


class FlatParams:
    """Flat fp32 parameter / gradient / optimizer-state buffers plus the Adam-WN plan.

    `shapes` are the logical (Keras) tensors; `aliases` maps some of them onto column slices of fused
    physical tensors (see fuse_heads)."""

    def __init__(self, shapes, device, phys=None, aliases=None, pre=0):
        """pre: floats of scratch IN FRONT of the gradient buffer, contiguous with it (`grads_pre`): what lives there is
        averaged across ranks together with the first gradient bucket (cl_vrnn: the optimizer's sum g.V of the hW kernel,
        which is linear in the gradient like the gradient itself)."""
        self.logical = list(shapes)
        self.aliases = dict(aliases or {})
        self.shapes = list(phys) if phys is not None else list(shapes)
        self.device = device
        self.offsets, self.col_offsets = {}, {}
        off = col = 0
        table = (_lib.ParamDesc * len(self.shapes))()
        for i, (name, shp) in enumerate(self.shapes):
            n = int(np.prod(shp))
            self.offsets[name] = off
            is_mat = len(shp) > 1
            rows = int(np.prod(shp[:-1])) if is_mat else 1
            table[i] = _lib.ParamDesc(off, rows, int(shp[-1]), col if is_mat else 0, int(is_mat), 0)
            if is_mat:
                self.col_offsets[name] = col
                col += (int(shp[-1]) + 3) // 4 * 4
            off += (n + 3) // 4 * 4
        self.n, self.n_cols, self.table = off, max(col, 4), table
        self._subplans = {}
        f = dict(dtype=torch.float32, device=device)
        self.params = torch.zeros(self.n, **f)
        pre = (int(pre) + 3) // 4 * 4
        self.grads_store = torch.zeros(pre + self.n, **f)
        self.grads_pre, self.grads = self.grads_store[:pre], self.grads_store[pre:]
        self.m = torch.zeros(self.n, **f)
        self.v = torch.zeros(self.n, **f)
        self.mg = torch.zeros(self.n_cols, **f)
        self.vg = torch.zeros(self.n_cols, **f)
        self.s = torch.ones(self.n_cols, **f)
        # ||V||^2 per column of the tall matrices, kept by every Adam-WN step (clv_adam_wn_step_ex): the next step's first
        # column sum.  norms_valid: it describes the parameters as they are now (cleared by anything else that writes them)
        self.vn2 = torch.zeros(self.n_cols, **f)
        self.norms_valid = False
        self.iterations = torch.zeros(1, dtype=torch.int32, device=device)
        L = _lib.lib()
        nb = L.clv_adam_wn_plan_bytes(table, len(self.shapes))
        blob = (C.c_uint8 * nb)()
        _lib.check(L.clv_adam_wn_plan_build(table, len(self.shapes), blob), "adam plan")
        self.plan = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(device)
        self.adam_ws = torch.empty(L.clv_adam_wn_workspace_bytes(table, len(self.shapes)), dtype=torch.uint8,
                                   device=device)

    # views ---------------------------------------------------------------
    def view(self, buf, name):
        if name in self.aliases:
            phys, c0, nc = self.aliases[name]
            return self.view(buf, phys)[..., c0:c0 + nc]
        shp = dict(self.shapes)[name]
        o = self.offsets[name]
        return buf[o:o + int(np.prod(shp))].view(*shp)

    def p(self, name):
        return self.view(self.params, name)

    def g(self, name):
        return self.view(self.grads, name)

    def rows(self, buf, name, r0):
        """1-D view of tensor `name` starting at row r0 (for sub-blocks of a kernel)."""
        shp = dict(self.shapes)[name]
        o = self.offsets[name] + r0 * int(shp[-1])
        return buf[o:]

    # host <-> device -------------------------------------------------------
    def set_weights(self, weights):
        for name, _ in self.logical:
            self.p(name).copy_(torch.as_tensor(np.asarray(weights[name], dtype=np.float32)))
        self.norms_valid = False

    def get_weights(self, buf=None):
        buf = self.params if buf is None else buf
        return {name: self.view(buf, name).detach().cpu().numpy().copy() for name, _ in self.logical}

    def state_tensors(self):
        """Everything a replica must share to step identically: parameters, Adam moments, the weight-norm column state
        and the step counter (which also keys the Philox noise stream)."""
        return [self.params, self.m, self.v, self.mg, self.vg, self.s, self.vn2, self.iterations]

    def reset_optimizer(self):
        for t in (self.m, self.v, self.mg, self.vg):
            t.zero_()
        self.s.fill_(1.0)
        self.iterations.zero_()
        self.norms_valid = False

    def _subplan(self, names):
        """(table, n, device plan) of the update restricted to the tensors in `names` (same flat buffers)."""
        key = tuple(names)
        if key not in self._subplans:
            idx = [i for i, (name, _) in enumerate(self.shapes) if name in names]
            table = (_lib.ParamDesc * len(idx))(*[self.table[i] for i in idx])
            L = _lib.lib()
            blob = (C.c_uint8 * L.clv_adam_wn_plan_bytes(table, len(idx)))()
            _lib.check(L.clv_adam_wn_plan_build(table, len(idx), blob), "adam plan")
            plan = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).to(self.device)
            self._subplans[key] = (table, len(idx), plan)
        return self._subplans[key]

    def tall_tensor(self):
        """Index and name of the one matrix of more than 144 rows (csrc/optim.hip SM_ROWS; cl_vrnn's hW/kernel), or None."""
        tall = [(i, name) for i, (name, shp) in enumerate(self.shapes) if len(shp) > 1 and int(np.prod(shp[:-1])) > 144]
        return tall[0] if len(tall) == 1 else None

    def adam_step(self, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, weightnorm=True, only=None, advance=True, advanced=False,
                  gdot=None):
        """utils/weightnorm.py:75-143; t comes from the device `iterations` counter.
        only: names of the tensors to update (default all); advance=False leaves `iterations` alone, so one optimizer step
        can be issued in pieces (the multi-GPU schedule updates a bucket as soon as its all-reduce has landed).
        advanced=True: the counter was already advanced by the launch that produced the gradients (the fused cl_vae
        step) and holds t.
        gdot [cols]: sum_j K[j,c] dK[j,c] of the tall matrix (ops.sparse_outer(gdot=...)), for gradients that were not
        averaged across ranks afterwards: with norms_valid the step takes the two-launch form (clv_adam_wn_step_ex)."""
        table, n, plan = (self.table, len(self.shapes), self.plan) if only is None else self._subplan(only)
        known, tall = None, self.tall_tensor()
        if int(weightnorm) == 1 and tall is not None and (only is None or tall[1] in only):
            use = gdot is not None and self.norms_valid
            pos = tall[0] if only is None else [name for name, _ in self.shapes if name in only].index(tall[1])
            known = _lib.AdamKnownSums(pos, int(use), ops._ptr(gdot) if use else None, ops._ptr(self.vn2))
        _lib.check(_lib.lib().clv_adam_wn_step_ex(
            table, n, ops._ptr(plan), ops._ptr(self.params), ops._ptr(self.grads),
            ops._ptr(self.m), ops._ptr(self.v), ops._ptr(self.mg), ops._ptr(self.vg), ops._ptr(self.s),
            ops._ptr(self.iterations), -2 if advanced else (0 if advance else -1), lr, b1, b2, eps, int(weightnorm),
            C.byref(known) if known is not None else None, ops._ptr(self.adam_ws),
            self.adam_ws.numel(), ops._stream()), "clv_adam_wn_step")
        # vn2 follows the parameters through whole Adam-WN steps only
        self.norms_valid = known is not None or (self.norms_valid and tall is not None and only is not None
                                                 and tall[1] not in only)


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
class VaeEngine(_EngineBase):
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

    def generate(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False, persistent=True, xhat_out=None):
        """N independent sequences of `nsteps` frames on the device: the frame loop of cl_vae/model.py:28-41
        (z-encoder on the last frame, z ~ N(mean, exp(lv)) or N(0, 1), decoder on (w, z, frame before last),
        x ~ Bernoulli); eps and u come from the Philox streams 0 / 1 at step = frame index.  x_seed [N,D], w [N,C] device
        tensors.  persistent=True (default where the shapes allow): the whole loop is ONE kernel, a workgroup per
        sequence (csrc/vae_generate.hip; any N); otherwise the layer chain captured once as a hipGraph and replayed per
        frame (N <= batch size).  Same noise, same samples either way."""
        cfg, d = self.cfg, self.device
        N, D, L = int(x_seed.shape[0]), cfg['D'], cfg['L']
        if persistent and cfg['H'] > 0 and ops.vae_generate_supported(D, cfg['H'], L, cfg['C']):
            P = self.P
            f = dict(dtype=torch.float32, device=d)
            Xs = torch.zeros(N, nsteps, D, **f)
            ops.vae_generate(N, nsteps, D, cfg['H'], L, cfg['C'], cfg['use_x_prev'], z_prior, seed,
                             x_seed.to(**f).contiguous(), w.to(**f).contiguous(), P.p('h/kernel'), P.p('h/bias'),
                             P.p('zargs/kernel'), P.p('zargs/bias'), P.p('decoder_h/kernel'), P.p('decoder_h/bias'),
                             P.p('x_decoded_mean/kernel'), P.p('x_decoded_mean/bias'), Xs, xhat_out)
            return Xs
        if N > self.B:
            raise ValueError("%d sequences exceed the engine's batch size %d" % (N, self.B))
        f = dict(dtype=torch.float32, device=d)
        x_in, hist, x_next = x_seed.to(**f).clone(), x_seed.to(**f).clone(), torch.zeros(N, D, **f)
        eps, u = torch.zeros(N, L, **f), torch.zeros(N, D, **f)
        counter = torch.zeros(1, dtype=torch.int32, device=d)
        Xs = torch.zeros(N, nsteps, D, **f)
        w = w.to(**f).contiguous()

        def frame():
            self.encode_z(x_in, w, N)
            ops.philox_normal(eps, N * L, seed, 0, 0, 0, step_dev=counter)
            if z_prior:
                self.zargs[:N].zero_()
            ops.gauss_fwd(N, L, self.zargs, eps, self.z, L, None)
            self.decode(w, self.z, hist if cfg['use_x_prev'] else None, N, act=ACT_SIGMOID)
            ops.philox_uniform(u, N * D, seed, 0, 1, 0, step_dev=counter)
            ops.bernoulli_sample(N * D, self.logits, u, x_next)
            ops.i32_add(counter, 1)
            hist.copy_(x_in)            # the decoder's history lags the encoder input by one frame
            x_in.copy_(x_next)

        graph = None
        for t in range(nsteps):
            if use_graph and t == 1:
                with ops.Graph() as graph:       # frame 0 ran eagerly and sized every workspace
                    frame()
            if graph is not None:
                graph.launch()
            else:
                frame()
            Xs[:, t].copy_(x_next)
        return Xs

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
        if stage is not None and target is None:
            # TrainStep handed over the mini-batch assembly (ops.label_stage): the kernel reads its rows' byte frames itself
            _lib.check(_lib.lib().clv_vae_fused_step_staged(
                B, cfg['D'], cfg['H'], cfg['Hc'], cfg['C'], cfg['L'], int(cfg['use_x_prev']), C.byref(stage), *tail),
                "clv_vae_fused_step_staged")
            return
        _lib.check(_lib.lib().clv_vae_fused_step_ex(
            B, cfg['D'], cfg['H'], cfg['Hc'], cfg['C'], cfg['L'], int(cfg['use_x_prev']), p_(x), p_(xp), p_(target),
            p_(w_true), *tail), "clv_vae_fused_step_ex")

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
class VrnnEngine(_EngineBase):
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
        self.fuse_xproj = bool(cfg.get('fuse_xproj', False)) and H == 88 and not self.dropout   # break-even vs the projection GEMM at config 3 (PERFLOG.md 8)
        # encoder + latent head + decoder as one launch (csrc/lstm_pair.hip); latent_dim <= 16
        self.fuse_pair = bool(cfg.get('fuse_pair', True)) and ops.lstm_pair_supported(L, H) and not self.fuse_xproj \
            and not self.dropout
        # Twice the workgroups, half the K each, for the LSTM weight-gradient products.  They run one 1024-thread
        # workgroup per CU with the register file full: alone on the GPU a grid of exactly 256 is best, but when the
        # gradient all-reduce holds a few CUs a 256-workgroup grid needs a whole second round; 512 shorter ones lose
        # only the share of the CUs that is taken.  TrainStep turns it on for world > 1.
        self.fine_grid = bool(cfg.get('fine_grid', os.environ.get('CLV_FINE_GRID', '0') == '1'))
        self.pair_pack = _f(d, ops.lstm_pair_pack_floats()) if self.fuse_pair else None
        # input projections by sparse row gathering (exact for any input; pays off for piano-roll frames)
        self.sparse_inputs = bool(cfg.get('sparse_inputs', True)) and ops.sparse_proj_supported(D, 4 * H)
        # output head: forward + loss + all three backward products in one launch
        self.fuse_head = bool(cfg.get('fuse_head', os.environ.get('CLV_FUSE_HEAD', '1') != '0')) \
            and ops.out_head_train_supported(H, D)
        self._head_done = False
        # the fused head also stores the logits (tests and callers of loss_and_grads read them); a replayed training step has
        # no reader for them, so TrainStep(use_graph=True) turns the store off: 11.5 MB per step at configuration 3, 92 MB at 5
        self.keep_logits = bool(cfg.get('keep_logits', True))
        # LSTM kernel gradients as split-bf16 products (6 of 9 piece pairs, <= 2^-25 per product: csrc/wgrad_bf16.hip) instead of the f32 MFMA GEMM;
        # frames_exact_bf16: every staged frame value is exactly a bf16 number (TrainStep sets it when the data set is
        # kept as uint8), which lets the frame rows use one bf16 piece instead of three
        self.bf16_wgrad = bool(cfg.get('bf16_wgrad', os.environ.get('CLV_BF16_WGRAD', '1') != '0'))
        self.frames_exact_bf16 = bool(cfg.get('frames_exact_bf16', False))
        # ... and both LSTMs' in one launch where they take the same form of the kernel (grads_tail)
        self.wgrad_pair = bool(cfg.get('wgrad_pair', os.environ.get('CLV_WGRAD_PAIR', '1') != '0'))
        # the hW layer's kernel gradient dense on the bf16 matrix cores when the frames are exact there (loss_and_grads)
        self.dense_hw_grad = bool(cfg.get('dense_hw_grad', os.environ.get('CLV_DENSE_HW_GRAD', '1') != '0'))
        # ... and its forward product (label forward): from CLV_DENSE_HW_FWD_ROWS batch rows on (default 512: at 256 rows the note-walking gather is as fast, 20.5 against 12.5 + 11.7 us)
        self.dense_hw_fwd = bool(cfg.get('dense_hw_fwd', os.environ.get('CLV_DENSE_HW_FWD', '1') != '0')) and \
            B >= int(os.environ.get('CLV_DENSE_HW_FWD_ROWS', '512'))
        self.ws_hw = None
        self.stage_spec = None       # set by TrainStep for one forward pass: see _forward_pair
        self._f8 = None              # the pass's uint8 frames (forward(frames8=...))
        self.ws_b = None
        # Note lists (opt-in: cfg['fuse_notes'] / CLV_FUSE_NOTES=1): when the batch was staged from BINARY uint8 frames,
        # the staging launch also writes each frame's list of notes (ops.gather_rows_multi(notes=...)) and the pair
        # forward kernel gathers the LSTM input projections itself -- no projection launch, no [B*T,352] round trip per
        # LSTM.  Measured (profiles/r03_notes_fusion_ab.txt): it LOSES on MI355X -- every CU re-gathers ~11 KB of kernel
        # rows per step through its L1 miss path (~10 B/cycle per CU), +48 us on the pair kernel for the 26 us launch it
        # removes -- so the default stays the projection launch, whose workgroups keep K_x in LDS.
        # notes_valid: the lists describe the frames now in X / XZ (TrainStep sets it per staged batch).
        # the label path's backward as the pair backward kernel's epilogue (CLV_LABEL_IN_PAIR=0: its own launch)
        self.label_in_pair = bool(cfg.get('label_in_pair', os.environ.get('CLV_LABEL_IN_PAIR', '1') != '0')) and self.fuse_pair
        # outside the pair kernels: the latent head's forward / backward as one MFMA launch each (csrc/latent_head.hip)
        self.fuse_latent = bool(cfg.get('fuse_latent', os.environ.get('CLV_FUSE_LATENT', '1') != '0')) \
            and ops.latent_head_supported(H, L)
        # large batches (>= 768 rows per GPU: BASELINE configuration 5) outside the pair kernels: both LSTMs' training passes
        # on the bf16 matrix cores with the frame rows of their input kernels gathered inside the kernel (csrc/lstm_mx.hip):
        # no projection launch, no [B*T,4H] projection buffer; gates_* / cs_* then hold the coefficient format
        self.use_mx = bool(cfg.get('lstm_mx', os.environ.get('CLV_USE_MX', '1') != '0')) and not self.fuse_pair \
            and not self.dropout and self.sparse_inputs and not self.fuse_xproj and ops.lstm_mx_supported(B, D, L, H)
        self.fuse_notes = bool(cfg.get('fuse_notes', os.environ.get('CLV_FUSE_NOTES', '0') == '1')) and self.fuse_pair \
            and self.sparse_inputs and D == ops.NOTE_NONE
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
            self.xm_e, self.xm_d = _f(d, 4, BT, D), _f(d, 4, BT, self.xz_ld)       # a gate's masked per-step inputs
            self.wm_e, self.wm_d = _f(d, 4, B, Cn), _f(d, 4, B, Cn)                # ... and masked label rows
            self.dxg, self.dwg = _f(d, BT, self.xz_ld), _f(d, B, Cn)               # a gate's share of dL/d[Xp | Z], dL/dW

    def set_dropout_uniforms(self, u_enc, u_dec):
        """The uniforms behind the two LSTMs' input-dropout masks ([B, 4, D + C] and [B, 4, (D) + L + C], m = (u >= p) / (1 - p)) as
        explicit inputs, like eps_W / eps_Z: a training pass without `noise=` uses them."""
        self.u_enc.copy_(u_enc.view_as(self.u_enc)); self.u_dec.copy_(u_dec.view_as(self.u_dec))
        self._masks_given = True

    def folds_noise(self):
        """True when forward(noise=...) draws eps_W / eps_Z inside the label kernel and the pair kernels / the latent head of the
        large-batch path (no Philox launch)."""
        return bool((self.fuse_pair or (self.use_mx and self.fuse_latent)) and self.sparse_inputs and self.cfg['D'] % 2 == 0)

    def frames_u8_supported(self):
        """Can a training pass read its frames as BYTES (frames8 of forward / loss_and_grads / grads_tail: the uint8 batch the
        staging launch copied out of the frame store, never widened to float)?  The large-batch path with every consumer of
        frames on its byte-reading kernel: csrc/lstm_mx.hip (note lists), wgrad_bf16.hip (x rows), out_head_bf16.hip (targets),
        outer_bf16.hip (the hW layer's forward and kernel-gradient products)."""
        cfg, B = self.cfg, self.B
        D, H, L, T = cfg['D'], cfg['H'], cfg['L'], cfg['T']
        if not (self.use_mx and self.fuse_head and self.bf16_wgrad and self.dense_hw_grad and self.dense_hw_fwd and D % 4 == 0
                and os.environ.get('CLV_FRAMES_U8', '1') != '0'):
            return False
        return bool(ops.dense_window_fwd_bf16_supported(B, T * D, D, T * D, D) and ops.dense_outer_bf16_supported(B, T * D, D, T * D, D)
                    and ops.sparse_dense_supported(D) and ops.lstm_wgrad_supported(4 * H, D, H, 0, 2)
                    and (not self.off or ops.lstm_wgrad_supported(4 * H, D, H, L, 2)))

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
        fuse_enc = self.fuse_xproj and ops.lstm_fused_input_fits(B, D)
        fuse_dec = self.fuse_xproj and ops.lstm_fused_input_fits(B, self.off + L)
        if not fuse_enc:
            if self.sparse_inputs:     # add the kernel rows of the notes that are on (csrc/sparse_proj.hip)
                ops.sparse_proj(BT, D, G4, X, D, P.p('encoder_h/kernel'), self.gates_enc)
            else:
                with self._side():
                    g(X, P.p('encoder_h/kernel'), self.gates_enc, BT, G4, D, ws=self.ws2)
        # label path (:174-191)
        off = self.off
        self._label_forward(X, eps_W, w_true)
        self._join()
        if fuse_enc:          # x_t.K gathered from the LDS-resident kernel inside the sequence kernel
            ops.lstm_seq_fwd_x(B, T, X, D, D, P.p('encoder_h/kernel'), self.wk_enc, P.p('encoder_h/recurrent_kernel'),
                               self.hs_enc, self.cs_enc, self.gates_enc, gate_act=self.gate_act)
        else:
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
        if fuse_dec:
            ops.lstm_seq_fwd_x(B, T, self.XZ, self.xz_ld, off + L, P.p('decoder_h/kernel'), self.wk_dec,
                               P.p('decoder_h/recurrent_kernel'), self.hs_dec, self.cs_dec, self.gates_dec,
                               gate_act=self.gate_act)
        elif off and self.sparse_inputs and ops.lstm_seq_fwd_z_supported(B, L, H):
            # large batches: the history frames' projection as a row gather, z_t . K_z inside the MFMA sequence kernel
            # (no dense [B*T, 120] x [120, 352] product, no second trip of the gate buffer through HBM)
            ops.sparse_proj(BT, off, G4, self.XZ, self.xz_ld, P.p('decoder_h/kernel'), self.gates_dec)
            ops.lstm_seq_fwd_z(B, T, self.gates_dec, self.wk_dec, P.p('decoder_h/recurrent_kernel'), self.Z, self.xz_ld, L,
                               P.rows(P.params, 'decoder_h/kernel', off), self.hs_dec, self.cs_dec, self.gates_dec,
                               gate_act=self.gate_act)
        else:
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

    # -- LSTM(dropout=p), training passes (cl_vrnn/model.py:164,198,227) ---------------------------------------------------
    # Keras 2.0.0 (implementation 0) multiplies the inputs of gate g's projection with a mask m_g [B, input_dim] drawn once per
    # batch -- the same for every time step -- and different for the four gates: z_g = ([x_t, z_t, W] * m_g) . K[:, g] + b_g +
    # h_{t-1} . U[:, g].  u_enc / u_dec [B, 4, input_dim] hold the masks' UNIFORMS (m = (u >= p) / (1 - p), applied by
    # clv_dropout_rows); the per-step rows and the label row of a gate are masked into xm_* / wm_* and multiplied as GEMMs
    # over that gate's column block (ldb = ldc = 4H).  Inference passes (validation, predict, generation) take no dropout.
    def _gate_cols(self, t, gi):
        """1-D / 2-D view of buffer t starting at gate gi's column block (pointer offset; the leading dimension stays 4H)"""
        H = self.cfg['H']
        return t[gi * H:] if t.dim() == 1 else t[:, gi * H:]

    def _forward_dropout(self, X, eps_W, eps_Z, w_true, nll, target):
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C'], self.off
        BT, G4, rate = B * T, 4 * H, self.dropout
        g, ws, gc = ops.gemm, self.ws, self._gate_cols
        self._label_forward(X, eps_W, w_true)              # hW, Wargs, W, the label losses (its unmasked row biases are replaced)
        in_e, in_d = D + Cn, off + L + Cn
        ue, ud = self.u_enc.view(B, 4 * in_e), self.u_dec.view(B, 4 * in_d)
        X2 = X.reshape(BT, D)
        for gi in range(4):
            ops.dropout_rows(BT, T, D, X2, D, ue[:, gi * in_e:], 4 * in_e, rate, self.xm_e[gi], D)
            g(self.xm_e[gi], gc(P.p('encoder_h/kernel'), gi), gc(self.gates_enc, gi), BT, H, D, ldb=G4, ldc=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.W, Cn, ue[:, gi * in_e + D:], 4 * in_e, rate, self.wm_e[gi], Cn)
            g(self.wm_e[gi], gc(P.rows(P.params, 'encoder_h/kernel', D), gi), gc(self.wk_enc, gi), B, H, Cn, ldb=G4, ldc=G4,
              bias=gc(P.p('encoder_h/bias'), gi), ws=ws)
        ops.lstm_seq_fwd(B, T, self.gates_enc, self.wk_enc, P.p('encoder_h/recurrent_kernel'), self.hs_enc, self.cs_enc,
                         self.gates_enc, gate_act=self.gate_act, H=H)
        if self.fuse_latent:
            ops.latent_head_fwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), P.p('Zargs/bias'), eps_Z, self.zargs, self.Z,
                                self.xz_ld, self.rowkl)
        else:
            g(self.hs_enc, P.p('Zargs/kernel'), self.zargs, BT, 2 * L, H, bias=P.p('Zargs/bias'), ws=ws)
            ops.gauss_fwd(BT, L, self.zargs, eps_Z, self.Z, self.xz_ld, self.rowkl)
        for gi in range(4):
            ops.dropout_rows(BT, T, off + L, self.XZ, self.xz_ld, ud[:, gi * in_d:], 4 * in_d, rate, self.xm_d[gi], self.xz_ld)
            g(self.xm_d[gi], gc(P.p('decoder_h/kernel'), gi), gc(self.gates_dec, gi), BT, H, off + L, lda=self.xz_ld, ldb=G4,
              ldc=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.W, Cn, ud[:, gi * in_d + off + L:], 4 * in_d, rate, self.wm_d[gi], Cn)
            g(self.wm_d[gi], gc(P.rows(P.params, 'decoder_h/kernel', off + L), gi), gc(self.wk_dec, gi), B, H, Cn, ldb=G4,
              ldc=G4, bias=gc(P.p('decoder_h/bias'), gi), ws=ws)
        ops.lstm_seq_fwd(B, T, self.gates_dec, self.wk_dec, P.p('decoder_h/recurrent_kernel'), self.hs_dec, self.cs_dec,
                         self.gates_dec, gate_act=self.gate_act, H=H)
        self._output_head(target, nll)

    def _backward_dropout(self, X, w_true, eps_W, eps_Z):
        """Everything behind dL/dh_dec (self.dhs) of a training pass with input dropout: both BPTTs, dZ and dW through the
        gates' masks, the latent head, the label path from an explicit dL/dW, every weight gradient.  Plain launches (no
        deferred reductions): this chain is the rarely used one."""
        cfg, P, B = self.cfg, self.P, self.B
        D, H, L, T, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C'], self.off
        C1, BT, G4, rate = Cn - 1, B * T, 4 * H, self.dropout
        g, ws, gc = ops.gemm, self.ws, self._gate_cols
        in_e, in_d = D + Cn, off + L + Cn
        ue, ud = self.u_enc.view(B, 4 * in_e), self.u_dec.view(B, 4 * in_d)
        Kd, Ke = P.p('decoder_h/kernel'), P.p('encoder_h/kernel')
        # decoder BPTT; dZ = sum_g m_g[z cols] * (dz_g . K_z[:, g]^T); dW (decoder share) likewise from sum_t dz
        ops.lstm_seq_bwd(B, T, P.p('decoder_h/recurrent_kernel'), self.dhs, self.cs_dec, self.gates_dec, self.dzsum_dec,
                         gate_act=self.gate_act, H=H)
        for gi in range(4):
            g(gc(self.gates_dec, gi), gc(P.rows(P.params, 'decoder_h/kernel', off), gi), self.dxg, BT, L, H, tb=True, lda=G4,
              ldb=G4, ldc=self.xz_ld, ws=ws)
            ops.dropout_rows(BT, T, L, self.dxg, self.xz_ld, ud[:, gi * in_d + off:], 4 * in_d, rate, self.dZ, L, beta=float(gi > 0))
            g(gc(self.dzsum_dec, gi), gc(P.rows(P.params, 'decoder_h/kernel', off + L), gi), self.dwg, B, Cn, H, tb=True, lda=G4,
              ldb=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.dwg, Cn, ud[:, gi * in_d + off + L:], 4 * in_d, rate, self.dW, Cn, beta=float(gi > 0))
        # latent head backward -> dL/dh_enc
        if self.fuse_latent:
            ops.latent_head_bwd(BT, H, L, self.hs_enc, P.p('Zargs/kernel'), self.zargs, eps_Z, self.dZ, L,
                                self.kl_weight / BT, self.dhs, P.g('Zargs/kernel'), P.g('Zargs/bias'), ws, defer=None)
        else:
            ops.gauss_bwd(BT, L, self.zargs, eps_Z, self.dZ, L, self.kl_weight / BT, self.dzargs)
            g(self.dzargs, P.p('Zargs/kernel'), self.dhs, BT, H, 2 * L, tb=True, ws=ws)
            self._dense_wgrad('Zargs', self.hs_enc, H, H, 2 * L, BT, self.dzargs, ws, None)
        ops.lstm_seq_bwd(B, T, P.p('encoder_h/recurrent_kernel'), self.dhs, self.cs_enc, self.gates_enc, self.dzsum_enc,
                         gate_act=self.gate_act, H=H)
        for gi in range(4):
            g(gc(self.dzsum_enc, gi), gc(P.rows(P.params, 'encoder_h/kernel', D), gi), self.dwg, B, Cn, H, tb=True, lda=G4,
              ldb=G4, ws=ws)
            ops.dropout_rows(B, 1, Cn, self.dwg, Cn, ue[:, gi * in_e + D:], 4 * in_e, rate, self.dW, Cn, beta=1.0)
        # label path from dL/dW (the LSTMs' share so far; clv_label_bwd adds the label losses' own terms)
        ops.label_bwd(B, Cn, self.wargs, self.wargs[:, C1:], 2 * C1, eps_W, w_true, self.W, self.dW,
                      cfg['w_log_var_prior'], self.class_weight, self.w_kl_weight, 1.0 / B,
                      self.dwargs, self.dwargs[:, C1:], 2 * C1)
        g(self.hW, self.dwargs, P.g('Wargs/kernel'), D, 2 * C1, B, ta=True, ws=ws)
        ops.colsum(self.dwargs, B, 2 * C1, P.g('Wargs/bias'), ws)
        g(self.dwargs, P.p('Wargs/kernel'), self.dhW, B, D, 2 * C1, tb=True, act=ACT_MASKPOS, aux=self.hW, ws=ws)
        g(X.reshape(B, T * D), self.dhW, P.g('hW/kernel'), T * D, D, B, ta=True, ws=ws)
        ops.colsum(self.dhW, B, D, P.g('hW/bias'), ws)
        self.gdot_fresh = False
        # weight gradients of the two LSTMs: per gate the masked inputs' products, the recurrent kernel as usual
        for name, xm, wm, nin, ldx, hs, dz, dzsum in (('encoder_h', self.xm_e, self.wm_e, D, D, self.hs_enc, self.gates_enc, self.dzsum_enc),
                                                      ('decoder_h', self.xm_d, self.wm_d, off + L, self.xz_ld, self.hs_dec, self.gates_dec,
                                                       self.dzsum_dec)):
            for gi in range(4):
                g(xm[gi], gc(dz, gi), gc(P.g(name + '/kernel'), gi), nin, H, BT, ta=True, lda=ldx, ldb=G4, ldc=G4, ws=ws)
                g(wm[gi], gc(dzsum, gi), gc(P.rows(P.grads, name + '/kernel', nin), gi), Cn, H, B, ta=True, ldb=G4, ldc=G4, ws=ws)
            ops.gemm_grouped_tn([dict(A=hs, lda=H, M=H, C=P.g(name + '/recurrent_kernel'), shift=1, zero_period=T)], G4, BT, dz, ws)
            ops.colsum(dzsum, B, G4, P.g(name + '/bias'), ws)
        if not self._head_done:
            self._dense_wgrad('X_decoded_mean', self.hs_dec, H, H, D, BT, self.dlogits, ws, None)
        rq = self._rq()
        if rq is not None:       # (the fused output head may have left its slabs pending; the loss means ride along)
            rq.flush(means=getattr(self, '_loss_terms', None), out=self.scal, skinny=None)
            self._loss_terms = None

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

    def _label_forward(self, X, eps_W, w_true, pack=None, stage=None):
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
                                 noise=nz[0] if nz else None, pack=pack, parts=parts, stage=stage)
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
        if stage is not None:
            self._label_forward(X, eps_W, w_true, pack=pack, stage=stage)
        notes = None
        if self.fuse_notes and self.notes_valid:      # the projections are gathered inside the pair kernel
            notes = (self.notes_enc, P.p('encoder_h/kernel'), self.notes_dec if off else None,
                     P.p('decoder_h/kernel') if off else None)
        elif self.sparse_inputs:     # piano-roll frames are ~4 % nonzero: add the kernel rows of the notes that are on
            if off:        # both LSTMs in one launch
                ops.sparse_proj2(BT, G4, (D, X, D, P.p('encoder_h/kernel'), self.gates_enc),
                                 (off, self.XZ, self.xz_ld, P.p('decoder_h/kernel'), self.gates_dec))
            else:
                ops.sparse_proj(BT, D, G4, X, D, P.p('encoder_h/kernel'), self.gates_enc)
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
        self._output_head(X if target is None else target, nll)

    def xp_view(self):
        """[B,T,D] strided view of the history columns of the [Xp | Z] buffer (stage batches straight into it)."""
        return self.XZ.view(self.B, self.cfg['T'], self.xz_ld)[:, :, :self.cfg['D']]

    def x_hat(self):
        """sigmoid(logits) of the last forward -> self.dlogits (the Keras output `X_decoded_mean`)."""
        cfg, P = self.cfg, self.P
        ops.gemm(self.hs_dec, P.p('X_decoded_mean/kernel'), self.dlogits, self.B * cfg['T'], cfg['D'], cfg['H'],
                 bias=P.p('X_decoded_mean/bias'), act=ACT_SIGMOID, ws=self.ws)
        return self.dlogits

    # -- stateful single-step inference (the reference's stateful batch-1 sub-models,
    #    cl_vrnn/model.py:116-162; here for any batch of independent sequences) -------------
    def new_state(self, B):
        d, H = self.device, self.cfg['H']
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=d)
        return dict(B=B, h_enc=z(B, H), c_enc=z(B, H), h_dec=z(B, H), c_dec=z(B, H), gates=z(B, 4 * H),
                    hs=z(B, H), zargs=z(B, 2 * self.cfg['L']), xhat=z(B, self.cfg['D']))

    def encode_w(self, X, B):
        """hW -> Wargs for B windows [B, T*D] (:174-181) -> self.wargs[:B]"""
        cfg, P = self.cfg, self.P
        D, T, C1 = cfg['D'], cfg['T'], cfg['C'] - 1
        ops.gemm(X, P.p('hW/kernel'), self.hW, B, D, T * D, bias=P.p('hW/bias'), act=ACT_RELU, ws=self.ws)
        ops.gemm(self.hW, P.p('Wargs/kernel'), self.wargs, B, 2 * C1, D, bias=P.p('Wargs/bias'), ws=self.ws)

    def _lstm_step(self, name, st, hkey, ckey):
        ops.lstm_seq_fwd(st['B'], 1, st['gates'], None, self.P.p(name + '/recurrent_kernel'), st['hs'], None, None,
                         h0=st[hkey], c0=st[ckey], hT=st[hkey], cT=st[ckey], gate_act=self.gate_act, H=self.cfg['H'])

    def enc_step(self, x, w, st, rec_name='encoder_h'):
        """one encoder-LSTM step on [x_t, w] + the Z heads -> st['zargs'] = [z_mean | z_log_var]"""
        cfg, P, B = self.cfg, self.P, st['B']
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g, ws = ops.gemm, self.ws
        g(x, P.p(rec_name + '/kernel'), st['gates'], B, 4 * H, D, ws=ws)
        g(w, P.rows(P.params, rec_name + '/kernel', D), st['gates'], B, 4 * H, Cn, beta=1.0, bias=P.p(rec_name + '/bias'),
          ws=ws)
        self._lstm_step(rec_name, st, 'h_enc', 'c_enc')
        g(st['hs'], P.p('Zargs/kernel'), st['zargs'], B, 2 * L, H, bias=P.p('Zargs/bias'), ws=ws)

    def dec_step(self, z, xp, w, st):
        """one decoder-LSTM step on [x_{t-1}, z_t, w] + sigmoid head -> st['xhat']"""
        cfg, P, B = self.cfg, self.P, st['B']
        D, H, L, Cn = cfg['D'], cfg['H'], cfg['L'], cfg['C']
        g, ws, off = ops.gemm, self.ws, self.off
        if cfg['use_x_prev']:
            g(xp, P.p('decoder_h/kernel'), st['gates'], B, 4 * H, D, ws=ws)
        g(z, P.rows(P.params, 'decoder_h/kernel', off), st['gates'], B, 4 * H, L,
          beta=1.0 if cfg['use_x_prev'] else 0.0, ws=ws)
        g(w, P.rows(P.params, 'decoder_h/kernel', off + L), st['gates'], B, 4 * H, Cn, beta=1.0,
          bias=P.p('decoder_h/bias'), ws=ws)     # three tiny GEMMs: batch-1 sampling is launch-bound, not flop-bound
        self._lstm_step('decoder_h', st, 'h_dec', 'c_dec')
        g(st['hs'], P.p('X_decoded_mean/kernel'), st['xhat'], B, D, H, bias=P.p('X_decoded_mean/bias'),
          act=ACT_SIGMOID, ws=ws)

    def generate(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False, persistent=True, xhat_out=None):
        """Autoregressive generation of N independent sequences on the device.  persistent=True (default where the
        shapes allow): the whole frame loop is ONE kernel, a workgroup per sequence (csrc/generate.hip); otherwise the
        per-frame chain below, captured once and replayed per frame.  Same Philox noise either way.
        xhat_out [N,S+nsteps,D] (persistent path only) receives every frame's note probabilities."""
        cfg = self.cfg
        if persistent and ops.vrnn_generate_supported(cfg['D'], cfg['H'], cfg['L'], cfg['C']):
            return self._generate_persistent(x_seed, w, nsteps, seed, z_prior, xhat_out)
        return self._generate_frames(x_seed, w, nsteps, seed, use_graph, z_prior)

    def _generate_persistent(self, x_seed, w, nsteps, seed, z_prior, xhat_out):
        cfg, P, d = self.cfg, self.P, self.device
        D, H, L, Cn, off = cfg['D'], cfg['H'], cfg['L'], cfg['C'], self.off
        N, S = int(x_seed.shape[0]), int(x_seed.shape[1])
        Xs = torch.zeros(N, nsteps, D, dtype=torch.float32, device=d)
        rows = lambda name, r: P.rows(P.params, name, r)
        ops.vrnn_generate(N, S, nsteps, D, H, L, Cn, self.gate_act, z_prior, seed, x_seed.contiguous() if S else None,
                          w.contiguous(), P.p('encoder_h/kernel'), rows('encoder_h/kernel', D), P.p('encoder_h/bias'),
                          P.p('encoder_h/recurrent_kernel'), P.p('Zargs/kernel'), P.p('Zargs/bias'),
                          P.p('decoder_h/kernel') if cfg['use_x_prev'] else None, rows('decoder_h/kernel', off),
                          rows('decoder_h/kernel', off + L), P.p('decoder_h/bias'), P.p('decoder_h/recurrent_kernel'),
                          P.p('X_decoded_mean/kernel'), P.p('X_decoded_mean/bias'), Xs, xhat_out)
        return Xs

    def _generate_frames(self, x_seed, w, nsteps, seed=0, use_graph=True, z_prior=False):
        """Batched autoregressive generation on the device (the hot loop of cl_vrnn/model.py:47-59 for N
        independent sequences at once, noise from Philox instead of np.random).
        x_seed [N,S,D] device tensor (teacher-forced frames, S may be 0), w [N,C]; returns Xs [N,nsteps,D].
        One frame = encoder step -> z ~ N(mean, exp(lv)) -> decoder step -> x ~ Bernoulli(x_hat); the chain
        is captured once and replayed per frame with no host synchronisation."""
        cfg, d = self.cfg, self.device
        N, S = int(x_seed.shape[0]), int(x_seed.shape[1])
        D, L = cfg['D'], cfg['L']
        f = dict(dtype=torch.float32, device=d)
        st = self.new_state(N)
        x_prev, x_next = torch.zeros(N, D, **f), torch.zeros(N, D, **f)
        eps, u, z = torch.zeros(N, L, **f), torch.zeros(N, D, **f), torch.zeros(N, L, **f)
        counter = torch.zeros(1, dtype=torch.int32, device=d)
        Xs = torch.zeros(N, nsteps, D, **f)
        w = w.contiguous()

        def frame():
            self.enc_step(x_prev, w, st)
            ops.philox_normal(eps, N * L, seed, 0, 0, 0, step_dev=counter)
            if z_prior:
                st['zargs'].zero_()
            ops.gauss_fwd(N, L, st['zargs'], eps, z, L, None)
            self.dec_step(z, x_prev if cfg['use_x_prev'] else None, w, st)
            ops.philox_uniform(u, N * D, seed, 0, 1, 0, step_dev=counter)
            ops.bernoulli_sample(N * D, st['xhat'], u, x_next)
            ops.i32_add(counter, 1)
            x_prev.copy_(x_next)

        if S == 0:
            x_prev.zero_()
        graph = None
        for t in range(S + nsteps):
            if t < S:
                x_prev.copy_(x_seed[:, t])
            if use_graph and t == 1:
                with ops.Graph() as graph:       # step 0 ran eagerly and sized every workspace
                    frame()
            if graph is not None:
                graph.launch()
            else:
                frame()
            if t >= S:
                Xs[:, t - S].copy_(x_next)
        return Xs

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
        elif H == 88 and L <= 40 and os.environ.get('CLV_BWD_Z', '1') != '0':      # dZ = dz_dec . Kz^T by two more waves of the decoder's backward kernel
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
