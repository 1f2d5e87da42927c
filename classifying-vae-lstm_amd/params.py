"""The flat, HBM-resident parameter / gradient / optimizer-state buffers of an engine (engine.FlatParams) and the
Adam-with-weight-norm step over them (utils/weightnorm.py:75-178 of the reference -> csrc/optim.hip)."""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops


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
        # ||V||^2 per column of the tall matrices, kept by every Adam-WN step (clv_adam_wn_step): the next step's first
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
        averaged across ranks afterwards: with norms_valid the step takes the two-launch form (clv_adam_wn_step)."""
        table, n, plan = (self.table, len(self.shapes), self.plan) if only is None else self._subplan(only)
        known, tall = None, self.tall_tensor()
        if int(weightnorm) == 1 and tall is not None and (only is None or tall[1] in only):
            use = gdot is not None and self.norms_valid
            pos = tall[0] if only is None else [name for name, _ in self.shapes if name in only].index(tall[1])
            known = _lib.AdamKnownSums(pos, int(use), ops._ptr(gdot) if use else None, ops._ptr(self.vn2))
        _lib.check(_lib.lib().clv_adam_wn_step(
            table, n, ops._ptr(plan), ops._ptr(self.params), ops._ptr(self.grads),
            ops._ptr(self.m), ops._ptr(self.v), ops._ptr(self.mg), ops._ptr(self.vg), ops._ptr(self.s),
            ops._ptr(self.iterations), -2 if advanced else (0 if advance else -1), lr, b1, b2, eps, int(weightnorm),
            C.byref(known) if known is not None else None, ops._ptr(self.adam_ws),
            self.adam_ws.numel(), ops._stream()), "clv_adam_wn_step")
        # vn2 follows the parameters through whole Adam-WN steps only
        self.norms_valid = known is not None or (self.norms_valid and tall is not None and only is not None
                                                 and tall[1] not in only)
