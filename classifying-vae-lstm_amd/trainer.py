"""One optimisation step of cl_vrnn / cl_vae on the HIP path, graph-captured.

Replaces the per-batch body of Keras Model.fit() at cl_vae/train.py:66-71 and
cl_vrnn/train.py:66-71: noise draw -> forward -> 4 losses -> backward ->
(data-parallel gradient average) -> Adam-with-weight-norm update.

Everything step-dependent lives on the device (Adam `iterations` counter, which also keys
the Philox noise stream), so the captured hipGraph is replayed unchanged every step; the
batch is staged into fixed buffers (`stage_batch`).
"""
import os

import numpy as np

import torch

from . import ops
from .parallel import GradAllReduce, eps_first_index, meta_device


class DevWindows:
    """Device side of utils.pianoroll.Windows: row i = store[starts[i] + t0 : ... + T] (store uint8 or float32 [F, D],
    starts int64 [n]); the batch gather reads the overlapping windows straight from the store."""

    def __init__(self, store, starts, t0):
        self.store, self.starts, self.t0 = store, starts, int(t0)
        self.shape = (int(starts.shape[0]),)


class TrainStep:
    def __init__(self, engine, seed=1234, rank=0, world=1, group=None, optimizer='adam-wn',
                 lr=1e-3, use_graph=True, fast_adam=True):
        self.eng = engine
        self.seed, self.rank, self.world = int(seed), int(rank), int(world)
        if optimizer not in ('adam-wn', 'adam', 'rmsprop'):
            raise ValueError("optimizer %r is not supported on the HIP path (adam-wn, adam, rmsprop)" % optimizer)
        self.weightnorm = {'adam': 0, 'adam-wn': 1, 'rmsprop': 2}[optimizer]      # CLV_OPT_* of include/clvae.h
        self.b2 = 0.9 if optimizer == 'rmsprop' else 0.999                       # rmsprop: rho
        self.lr = lr
        self.use_graph = use_graph
        # nothing reads the logits of a replayed training step (VrnnEngine.keep_logits): the store is switched off AROUND this
        # step's own passes (_main), never on the shared engine for good -- a direct loss_and_grads() call, a tool or a second
        # TrainStep(use_graph=False) on the same engine keep finding fresh logits
        self._drop_logits = bool(use_graph and hasattr(engine, 'keep_logits') and 'keep_logits' not in engine.cfg)
        self.fast_adam = bool(fast_adam) and os.environ.get('CLV_FAST_ADAM', '1') != '0'      # see _update()
        cfg, B, d = engine.cfg, engine.B, engine.device
        self.is_vrnn = 'T' in cfg
        T = cfg['T'] if self.is_vrnn else 1
        D, L, C1 = cfg['D'], cfg['L'], cfg['C'] - 1
        f = dict(dtype=torch.float32, device=d)
        shp = (B, T, D) if self.is_vrnn else (B, D)
        self.X = torch.zeros(*shp, **f)
        # cl_vrnn with history: stage x_{t-1} straight into the engine's [Xp | Z] decoder-input buffer
        self.xp_ld = 0
        if self.is_vrnn and cfg['use_x_prev']:
            self.Xp = engine.xp_view()
            self.xp_ld = engine.xz_ld
        else:
            self.Xp = torch.zeros(*shp, **f)
        self.Y = None        # reconstruction target when it is not the input itself (--predict_next): see set_target()
        # The bound batches of the large-batch path stay BYTES (round 6): the staging launch copies the rows out of the uint8 frame
        # store as they are (X8 / Xp8, a quarter of the float batch) and every kernel of the step reads frames from them
        # (VrnnEngine.frames_u8_supported); X / Xp are then not written at all.  _f8: the captured step uses them.
        self.X8 = self.Xp8 = None
        self._f8 = None
        self.w_true = torch.zeros(B, cfg['C'], **f)
        self.eps_w = torch.zeros(B, C1, **f)
        self.eps_z = torch.zeros(B * T, L, **f)
        tail = engine.tail_range() if self.is_vrnn else (0, 0)
        # CLV_FORCE_DP_GRAPHS=1: take the multi-GPU schedule (main graph, tail graph, update graph around the two
        # gradient buckets) on a single GPU too, with the collectives as no-ops -- lets one box test that path
        force = os.environ.get('CLV_FORCE_DP_GRAPHS') == '1'
        # the scratch in front of the gradients (cl_vrnn: the optimizer's sum g.V of the hW kernel) travels with the tail bucket
        P = engine.P
        self.pre_in_tail = bool(self.is_vrnn and tail[1] and tail[0] == 0 and P.grads_pre.numel() > 0
                                and getattr(engine, 'gdot', None) is not None
                                and engine.gdot.data_ptr() == P.grads_pre.data_ptr())
        # CLV_DP_REAL_COLLECTIVES=1 (with the forced schedule, inside a one-rank RCCL group): the collectives are issued
        # for real -- the only way a one-GPU box can run RCCL inside this schedule, and try to capture it
        real = os.environ.get('CLV_DP_REAL_COLLECTIVES') == '1'
        if self.pre_in_tail:
            self.ar = GradAllReduce(P.grads_store, 0, P.grads_pre.numel() + tail[1], group, always=real) \
                if (world > 1 or force) else None
        else:
            self.ar = GradAllReduce(P.grads, tail[0], tail[1], group, always=real) if (world > 1 or force) else None
        # the tail bucket is exactly one tensor (the hW kernel): its update can run while the main bucket is reduced
        self.tail_names = [n for n, _ in engine.P.shapes if tail[1] and engine.P.offsets[n] == tail[0]
                           and int(np.prod(dict(engine.P.shapes)[n])) == tail[1]]
        self.rest_names = [n for n, _ in engine.P.shapes if n not in self.tail_names]
        self.split_update = self.ar is not None and len(self.tail_names) == 1
        # The big bucket's all-reduce runs next to the LSTM weight-gradient products (VrnnEngine.fine_grid: twice the
        # workgroups, half the K each, so that a few CUs held by RCCL do not cost a whole second round).  Alone on a GPU
        # the fine grid is 8-15 % slower (profiles/r03_dp_schedule_one_gpu.txt) and nobody has measured it next to a real
        # all-reduce, so it is no longer assumed: tune_dp_schedule() times both on the hardware at hand and keeps the
        # faster (bench.py and Model.fit call it before their first step).  Until then: the coarse grid.
        self.dp_trials = None
        # One graph for the whole data-parallel step, collectives included, when RCCL lets itself be captured (tried once, at the
        # first capture; `capture_note` says what happened); else two graphs + plain launches around eager collectives.
        # DEFAULT since round 6 (CLV_CAPTURE_COLLECTIVES=0 keeps the split schedule): on the MI355X box's RCCL 2.26 the capture
        # works (one-rank group, the only group a one-GPU box can build) and the captured step takes 0.385 ms against 0.407 for
        # the split schedule around the same two real RCCL calls (0.353: the single-GPU step; profiles/r06_dp_schedule_one_gpu.txt).
        # A capture that raises falls back to the split schedule and says so; nobody has run either across GPUs yet.
        self.capture_collectives = os.environ.get('CLV_CAPTURE_COLLECTIVES', '1') != '0'
        self.capture_note = None
        self._graphs = None
        self._warm = False
        self._bound = None          # bind_batches(): the mini-batch assembly as the first node of the captured step
        self.loss_acc = None        # a [>= 5] float tensor: every step adds its five loss means to it (inside the graph)

    # -- pieces -----------------------------------------------------------
    def noise_spec(self, stream_offset=0, row0=None):
        """(seed, stream_w, stream_z, first_w, first_z, step, step_dev) of this rank's eps_w / eps_z: the Philox streams
        (2*stream_offset, 2*stream_offset+1) at its global rows (row0 = first global row of the batch held here; default
        rank * B), at step `iterations`."""
        eng, B = self.eng, self.eng.B
        C1, L = eng.cfg['C'] - 1, eng.cfg['L']
        T = eng.cfg['T'] if self.is_vrnn else 1
        row0 = self.rank * B if row0 is None else int(row0)      # global row of this rank's first sample
        return (self.seed, 2 * stream_offset, 2 * stream_offset + 1, eps_first_index(row0, C1),
                eps_first_index(row0, T * L), 0, eng.P.iterations)

    def draw_noise(self, stream_offset=0, row0=None):
        """Fill eps_w / eps_z (noise_spec) with one launch."""
        eng, B = self.eng, self.eng.B
        T = eng.cfg['T'] if self.is_vrnn else 1
        seed, sw, sz, fw, fz, step, it = self.noise_spec(stream_offset, row0)
        ops.philox_normal2(self.eps_w, B * (eng.cfg['C'] - 1), sw, fw, self.eps_z, B * T * eng.cfg['L'], sz, fz, seed, step,
                           step_dev=it)

    def _folded(self):
        """cl_vae's fused step draws its own noise and advances `iterations` in the launches it already has (a step is
        launch-bound: three launches fewer is a sixth of it)."""
        return (not self.is_vrnn) and self.eng.folds_step(self.w_true)

    def set_target(self, on):
        """on: the decoder output is scored against a separate target batch (staged next to the inputs) instead of the
        input frames.  Changes what the captured step reads, so the graphs are dropped."""
        if bool(on) != (self.Y is not None):
            self.Y = torch.zeros_like(self.X) if on else None
            self.recapture()

    def bind_batches(self, d_cur, d_hist, d_w, idx=None, period=1, stride=None, offset=0, d_target=None):
        """From now on step() assembles its own mini-batch, as the first node of the captured graph: batch j =
        (iterations - iterations at the first bound step) mod `period`, rows idx[j * stride + offset + r] (idx None: rows
        j * stride + offset + r) of the device-resident data set, r < B.  `iterations` is the optimizer's device counter,
        so nothing is staged from the host: a step is one graph launch (Model.fit: idx = the epoch's permutation, rewritten
        in place per epoch; period = batches per epoch; stride = global batch; offset = rank * B)."""
        self.set_target(d_target is not None)
        self._bound = dict(cur=d_cur, hist=d_hist, w=d_w, idx=idx, period=int(period),
                           stride=int(self.eng.B if stride is None else stride), offset=int(offset), target=d_target,
                           step0=int(self.eng.P.iterations.item()))       # (one host read, here: never inside a capture)
        self.recapture()

    def unbind_batches(self):
        if self._bound is not None:
            self._bound = None
            self.recapture()

    def _bytes_batch(self, b):
        """(X8, Xp8) when the bound batches can stay uint8 for the whole step, else None (see __init__)."""
        eng = self.eng
        if not (self.is_vrnn and b['target'] is None and getattr(eng, 'frames_u8_route', lambda: None)() == 'gather'):
            return None
        need_hist = eng.off > 0
        if need_hist != (b['hist'] is not None):
            return None
        for x in ((b['cur'], b['hist']) if need_hist else (b['cur'],)):
            t = x.store if isinstance(x, DevWindows) else x
            if t.dtype != torch.uint8 or not t.is_contiguous():
                return None
        if self.X8 is None:
            shp = tuple(self.X.shape)
            self.X8 = torch.zeros(*shp, dtype=torch.uint8, device=self.X.device)
            self.Xp8 = torch.zeros(*shp, dtype=torch.uint8, device=self.X.device) if need_hist else None
        return (self.X8, self.Xp8)

    def _stage_bound(self):
        b = self._bound
        self._f8 = self._bytes_batch(b)
        segs = self._segments(b['cur'], b['hist'], b['w'], b['target'], bytes_out=self._f8)
        ops.gather_rows_multi(self.eng.B, b['idx'], segs, notes=self._note_outputs(b['cur'], b['hist'], len(segs)),
                              cursor=(self.eng.P.iterations, b['step0'], b['period'], b['stride'], b['offset']))

    def _label_stage(self):
        """ops.label_stage of the bound batches when the step's first launch can assemble them itself -- cl_vrnn: the label
        forward launch (VrnnEngine.can_stage_in_label), cl_vae: the fused step kernel -- so that the mini-batch assembly is no
        launch of its own; else None.  Byte frames only, no separate target, no note lists; CLV_STAGE_IN_LABEL=0 keeps the
        gather launch."""
        b, eng = self._bound, self.eng
        if os.environ.get('CLV_STAGE_IN_LABEL', '1') == '0' or b['target'] is not None:
            return None
        if not getattr(eng, 'can_stage_in_label', lambda: False)():
            return None
        if not self.is_vrnn and not (self._folded() and b['w'] is not None):
            return None
        need_hist = (eng.off > 0) if self.is_vrnn else bool(eng.cfg['use_x_prev'])
        if need_hist != (b['hist'] is not None):
            return None
        D, row = eng.cfg['D'], int(self.X[0].numel())

        def src(x):
            if isinstance(x, DevWindows):
                return (x.store, D, x.t0 * D, x.starts)
            return (x, row, 0, None)
        cur = src(b['cur'])
        hist = src(b['hist']) if need_hist else None
        if any(t is not None and (t[0].dtype != torch.uint8 or not t[0].is_contiguous()) for t in (cur, hist)):
            return None
        if b['w'].dtype != torch.float32 or not b['w'].is_contiguous():
            return None
        if hasattr(eng, 'frames_exact_bf16') and not eng.frames_exact_bf16:      # byte frames (what _segments() notes for the gather)
            eng.frames_exact_bf16 = True
            self.recapture()
        hist_chunk, hist_ld = (D, self.xp_ld) if self.xp_ld else (row, row)
        f8 = None
        if self.is_vrnn and getattr(eng, 'frames_u8_route', lambda: None)() == 'label':
            # the label launch leaves the batch as BYTES (a quarter of its stores) and every later launch of the step reads bytes
            if self.X8 is None:
                self.X8 = torch.zeros(tuple(self.X.shape), dtype=torch.uint8, device=self.X.device)
                self.Xp8 = torch.zeros(tuple(self.X.shape), dtype=torch.uint8, device=self.X.device) if need_hist else None
            f8 = (self.X8, self.Xp8)
        self._f8 = f8
        return ops.label_stage(cur, hist, b['idx'], 0, (eng.P.iterations, b['step0'], b['period'], b['stride'], b['offset']),
                               self.X, self.Xp if need_hist else None, hist_chunk, hist_ld, b['w'], self.w_true, bytes_out=f8)

    def _main(self):
        if not self._drop_logits:
            return self._main_pass()
        keep, self.eng.keep_logits = self.eng.keep_logits, False
        try:
            return self._main_pass()
        finally:
            self.eng.keep_logits = keep

    def _main_pass(self):
        self._f8 = None
        if self._bound is not None:
            st = self._label_stage()
            if st is None:
                self._stage_bound()
            elif hasattr(self.eng, 'stage_spec'):
                self.eng.stage_spec = st
        if self._folded():
            self.eng.loss_and_grads(self.X, self.Xp, self.w_true, self.eps_w, self.eps_z, target=self.Y,
                                    noise=self.noise_spec(), bump=True)
            return
        if self.is_vrnn:     # eps is drawn inside the label / pair kernels where they run (else one Philox launch in forward())
            self.eng.loss_and_grads(self.X, self.Xp, self.w_true, self.eps_w, self.eps_z, do_tail=False, target=self.Y,
                                    noise=self.noise_spec(), frames8=self._f8)
            return
        self.draw_noise()
        self.eng.loss_and_grads(self.X, self.Xp, self.w_true, self.eps_w, self.eps_z, target=self.Y)

    def _tail(self):
        if self.is_vrnn:
            self.eng.grads_tail(self.X, frames8=self._f8)

    def _update(self):
        # single GPU, cl_vrnn: the backward pass left sum_j K dK of the hW kernel (VrnnEngine.gdot), so Adam-WN runs in two
        # launches instead of five (FlatParams.adam_step); not under data parallelism (the gradient is averaged afterwards)
        eng = self.eng
        eng.P.adam_step(lr=self.lr, b2=self.b2, weightnorm=self.weightnorm, advanced=self._folded(), gdot=self._gdot())

    def _gdot(self):
        """The hW kernel's sum g.V when this step's backward pass left it AND it went through whatever averaged the
        gradient (single GPU: nothing; data parallel: the tail bucket, which then carries it)."""
        eng = self.eng
        ok = self.fast_adam and getattr(eng, 'gdot_fresh', False) and (self.ar is None or self.pre_in_tail)
        if hasattr(eng, 'gdot_fresh'):
            eng.gdot_fresh = False
        return eng.gdot if ok else None

    # multi-GPU: the optimizer step in two pieces, the tail bucket's tensor first (its all-reduce has landed under
    # _tail()), the rest once the main bucket is in; `iterations` advances with the second piece
    def _update_tail(self):
        self.eng.P.adam_step(lr=self.lr, b2=self.b2, weightnorm=self.weightnorm, only=self.tail_names, advance=False,
                             advanced=self._folded(), gdot=self._gdot())

    def _update_rest(self):
        self.eng.P.adam_step(lr=self.lr, b2=self.b2, weightnorm=self.weightnorm, only=self.rest_names,
                             advanced=self._folded())

    # -- public -----------------------------------------------------------
    def _segments(self, cur, hist, w, target=None, bytes_out=None):
        """(src, out, row_elems, chunk, out_ld[, stride, offset, table]) of the current frames, history frames, labels and
        (optional) target frames of a batch.  cur / hist / target are device tensors of whole rows or DevWindows (windows
        of a frame store).  bytes_out = (X8, Xp8): the frames are copied as bytes into these uint8 buffers (_bytes_batch)."""
        row, D = int(self.X[0].numel()), self.eng.cfg['D']

        def src(x):
            return (x.store, (D, x.t0 * D, x.starts)) if isinstance(x, DevWindows) else (x, ())
        # frames that arrive as bytes are exactly representable in bf16: the LSTM kernel-gradient products then need
        # one bf16 piece for the frame rows (VrnnEngine.frames_exact_bf16); the choice is baked into the captured step
        exact = all(src(x)[0].dtype == torch.uint8 for x in (cur, hist) if x is not None)
        if hasattr(self.eng, 'frames_exact_bf16') and self.eng.frames_exact_bf16 != exact:
            self.eng.frames_exact_bf16 = exact
            self.recapture()
        c, cx = src(cur)
        if bytes_out is not None:
            segs = [(c, bytes_out[0], row, D, D) + cx]
            if hist is not None:
                h, hx = src(hist)
                segs.append((h, bytes_out[1], row, D, D) + hx)
            segs.append((w, self.w_true, int(self.w_true.shape[1]), 0, 0))
            return segs
        segs = [(c, self.X, row, D, D) + cx]          # frame by frame (the same bytes as one piece; note lists are per frame)
        if hist is not None:
            h, hx = src(hist)
            if self.xp_ld:      # history frames go straight into the [Xp | Z] decoder-input buffer
                segs.append((h, self.Xp, row, D, self.xp_ld) + hx)
            else:
                segs.append((h, self.Xp, row, 0, 0) + hx)
        segs.append((w, self.w_true, int(self.w_true.shape[1]), 0, 0))
        if target is not None:
            t, tx = src(target)
            segs.append((t, self.Y, row, 0, 0) + tx)
        return segs

    def _note_outputs(self, cur, hist, nseg):
        """Note-list outputs of the staging launch (VrnnEngine.fuse_notes): the current and the history frames get lists
        when both come from uint8 stores that hold only 0 / 1 (checked once per store: a list says which notes are on,
        not how loud).  Tells the engine whether its lists describe the staged batch (baked into the captured step)."""
        eng = self.eng
        if not getattr(eng, 'fuse_notes', False):
            return None
        need = [cur, hist] if eng.off else [cur]           # a decoder without history frames needs no list of them
        ok = all(x is not None and self._is_binary_u8(x.store if isinstance(x, DevWindows) else x) for x in need)
        if eng.notes_valid != ok:
            eng.notes_valid = ok
            self.recapture()
        if not ok:
            return None
        out = [None] * nseg                                # segments: current frames, [history frames], labels, [targets]
        out[0] = eng.notes_enc
        if eng.off:
            out[1] = eng.notes_dec
        return out

    def _is_binary_u8(self, t):
        """uint8 and every value 0 or 1; one reduction (and host sync) per distinct storage, remembered."""
        if t.dtype != torch.uint8:
            return False
        key = (t.untyped_storage().data_ptr(), t.untyped_storage().nbytes())
        cache = self.__dict__.setdefault('_binary_stores', {})
        if key not in cache:
            whole = torch.empty(0, dtype=torch.uint8, device=t.device).set_(t.untyped_storage())
            cache[key] = bool((whole <= 1).all().item())
        return cache[key]

    def stage_batch(self, X, Xp, w_true, target=None):
        """Copy one batch (contiguous device tensors) into the fixed staging buffers: one launch."""
        self.set_target(target is not None)
        segs = self._segments(X, Xp, w_true, target)
        ops.gather_rows_multi(self.eng.B, None, segs, notes=self._note_outputs(X, Xp, len(segs)))

    def gather_batch(self, d_cur, d_hist, d_w, ib, row0=0, d_target=None):
        """Assemble the batch rows `ib` (device int64 indices; None = rows row0..row0+B-1) from the HBM-resident
        data set (frames float32 or uint8): one launch."""
        self.set_target(d_target is not None)
        segs = self._segments(d_cur, d_hist, d_w, d_target)
        ops.gather_rows_multi(self.eng.B, ib, segs, row0=row0, notes=self._note_outputs(d_cur, d_hist, len(segs)))

    def _accumulate(self):
        if self.loss_acc is not None:
            ops.axpy(5, 1.0, self.eng.scal, self.loss_acc)

    def _single(self):
        """The whole step on one GPU (eager or under capture)."""
        self._main()
        self._tail()
        self._update()
        self._accumulate()

    def _eager(self):
        if self.ar is None:
            return self._single()
        self._main()
        if self.ar is not None:
            self.ar.reduce_tail()       # the hW-kernel bucket (most of the bytes) is complete: reduce it under _tail()
        self._tail()
        if self.ar is not None:
            self.ar.reduce_main()
            if self.split_update:
                self.ar.wait_tail()
                self._update_tail()
                self.ar.wait()
                self._update_rest()
                self._accumulate()
                return
            self.ar.wait()
        self._update()
        self._accumulate()

    def tune_dp_schedule(self, steps=20, warm=3):
        """Data parallel only: time `steps` steps of the staged batch with the coarse and the fine weight-gradient grid
        (VrnnEngine.fine_grid) on THIS hardware, next to the real collectives, and keep the faster; parameters and
        optimizer state are restored.  Every rank runs the same trials, the times are MAX-reduced, so every rank takes
        the same decision.  Returns (and keeps in `dp_trials`) {'coarse_ms', 'fine_ms', 'chosen'}."""
        import time
        eng = self.eng
        if self.ar is None or not hasattr(eng, 'fine_grid') or 'fine_grid' in eng.cfg:
            return None
        import torch.distributed as dist
        saved = [t.clone() for t in eng.P.state_tensors()]
        flags = (eng.P.norms_valid, getattr(eng, 'gdot_fresh', False))
        res = {}
        for name, fine in (('coarse_ms', False), ('fine_ms', True)):
            eng.fine_grid = fine
            self.recapture()
            self._warm = False             # like a fresh run: one eager step (it validates the column norms), then the capture
            for _ in range(warm):
                self.step()
            torch.cuda.synchronize()
            if self.ar.world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            torch.cuda.synchronize()
            ms = torch.tensor([1e3 * (time.perf_counter() - t0) / steps], dtype=torch.float64, device=meta_device(eng.device))
            if self.ar.world > 1:
                dist.all_reduce(ms, op=dist.ReduceOp.MAX)
            res[name] = round(float(ms.item()), 4)
            for t, sv in zip(eng.P.state_tensors(), saved):
                t.copy_(sv)
            eng.P.norms_valid = flags[0]
            if hasattr(eng, 'gdot_fresh'):
                eng.gdot_fresh = flags[1]
        eng.fine_grid = res['fine_ms'] < res['coarse_ms']
        res['chosen'] = 'fine' if eng.fine_grid else 'coarse'
        self.recapture()
        self._warm = False                 # the restored state has no valid norms: the next step is eager, then the capture
        self.dp_trials = res
        return res

    def recapture(self):
        """Drop the captured graphs (values baked into them changed, e.g. annealed loss weights); the next step()
        captures again.  Buffers, streams and the all-reduce buckets stay."""
        self._graphs = None

    def step(self):
        if not self.use_graph:
            self._eager()
            return
        if not self._warm:               # first call sizes every workspace; graphs come next
            self._eager()
            self._warm = True
            return
        if self._graphs is None:
            if self.ar is None:
                # the captured optimizer takes its short form only while the column norms follow the parameters
                # (FlatParams.norms_valid); a step after they were written from outside runs eagerly (and re-validates)
                self._graph_fast = bool(getattr(self.eng.P, 'norms_valid', False))
                with ops.Graph() as g:
                    self._single()
                self._graphs = (g,)
            else:
                self._graph_fast = bool(getattr(self.eng.P, 'norms_valid', False))
                if self._capture_whole_dp_step():
                    return self._launch_dp()
                with ops.Graph() as g1:
                    self._main()
                # whether the captured backward pass writes the hW kernel's sum g.V (only its sparse form does): the
                # replayed steps restore the flag from this, never assume it
                self._main_leaves_gdot = bool(getattr(self.eng, 'gdot_fresh', False))
                with ops.Graph() as g2:
                    self._tail()
                if self.split_update and os.environ.get('CLV_DP_EAGER_UPDATE', '1') != '0':
                    # the two optimizer pieces as plain launches: graphs of 1-3 kernels cost more than they save (forced
                    # schedule on one GPU: 0.4255 -> 0.4145 ms per step)
                    self._graphs = (g1, g2, None, None)
                elif self.split_update:
                    with ops.Graph() as g3:
                        self._update_tail()
                    with ops.Graph() as g4:
                        self._update_rest()
                    self._graphs = (g1, g2, g3, g4)
                else:
                    with ops.Graph() as g3:
                        self._update()
                    self._graphs = (g1, g2, g3)
        if self.ar is None:
            if getattr(self, '_graph_fast', False) and not self.eng.P.norms_valid:
                self._eager()
                return
            self._graphs[0].launch()
        else:
            self._launch_dp()

    def _capture_whole_dp_step(self):
        """Try ONE graph for the data-parallel step: backward (early part), bucket 1 all-reduce on the side stream,
        backward (late part), bucket 2, the two optimizer pieces.  RCCL (torch's ProcessGroupNCCL) under a foreign
        stream capture is not promised to work: any failure is caught, noted, and the split schedule is used."""
        import torch.distributed as dist
        rccl = self.ar is not None and self.ar.live and self.ar.cuda and dist.get_backend(self.ar.group) == 'nccl'
        if not (self.capture_collectives and rccl):
            if self.capture_note is None:
                self.capture_note = "not tried: " + ("the collectives of this process are not RCCL calls on device buffers"
                                                     if not rccl else "switched off (CLV_CAPTURE_COLLECTIVES=0)")
            return False
        if self.capture_note is not None and not self.capture_note.startswith('captured'):
            return False                     # failed before: do not try again
        try:
            with ops.Graph() as g:
                self._eager()
            self._main_leaves_gdot = False   # (the optimizer pieces are inside the graph: nothing reads the host flag)
            self._graphs = ('whole', g)
            self.capture_note = "captured: one graph per step, 2 all-reduces inside"
            return True
        except Exception as e:               # noqa: BLE001 -- whatever RCCL / HIP / torch raised during the capture
            self.capture_note = "capture failed, split schedule kept: %s" % (repr(e)[:300],)
            torch.cuda.synchronize()
            self.ar.side = torch.cuda.Stream(device=self.eng.device)     # the old side stream may still count as capturing
            return False

    def _launch_dp(self):
        if getattr(self, '_graph_fast', False) and not self.eng.P.norms_valid:
            self._eager()
            return
        if self._graphs[0] == 'whole':
            self._graphs[1].launch()
            return
        g1, g2, g3 = self._graphs[:3]
        g1.launch()
        self.ar.reduce_tail()       # hW-kernel bucket, overlaps the weight-gradient products of g2
        g2.launch()
        self.ar.reduce_main()
        if self.split_update:
            self.ar.wait_tail()
            # Adam-WN of the hW kernel (87 % of the parameters) under the main bucket's all-reduce
            if g3 is not None:
                g3.launch()
            else:
                if hasattr(self.eng, 'gdot_fresh'):
                    self.eng.gdot_fresh = self._main_leaves_gdot      # what the replayed backward pass has left
                self._update_tail()
            self.ar.wait()
            if self._graphs[3] is not None:
                self._graphs[3].launch()
            else:
                self._update_rest()
        else:
            self.ar.wait()
            g3.launch()
        self._accumulate()
