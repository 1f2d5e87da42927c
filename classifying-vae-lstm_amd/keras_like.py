"""The slice of the Keras 2.0.0 `Model` surface that the reference's scripts use, on the HIP engines.

The reference calls: fit(x, y, shuffle, epochs, batch_size, callbacks,
validation_data) -> history with .history[str] -> list (cl_vae/train.py:66-73,
cl_vrnn/train.py:66-73); predict(x | [x...]) (cl_vae/model.py:25,29,38; cl_vrnn/model.py:40,50,57);
save_weights / load_weights (utils/model_utils.py:138; cl_vae/model.py:238); to_yaml
(utils/model_utils.py:164); get_layer(name).get_weights()/.set_weights() (cl_vrnn/model.py:130-133,
160-161); reset_states() (cl_vrnn/model.py:22-24); .layers, .inputs, .stop_training (callbacks).
Semantics of fit() follow SURVEY.md Appendix A.4 (Keras 2.0.0, recalled).

Nothing here computes on the host: every batch is a replay of the captured training step
(trainer.TrainStep) on HBM-resident data; the host only shuffles indices and keeps the books.
"""
import json

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, ops
from .trainer import TrainStep
from .utils import h5io


class Variable:
    """keras.backend.variable for the annealed loss weights (cl_vae/train.py:42,48)."""

    def __init__(self, value, name=None):
        self.value = float(value)
        self.name = name


def get_value(v):
    return v.value if isinstance(v, Variable) else float(v)


def set_value(v, x):
    v.value = float(x)


class Callback:
    """keras.callbacks.Callback protocol used by utils/model_utils.py."""
    rank0_only = False       # True for callbacks that write files: data-parallel runs keep them on rank 0

    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class Layer:
    """A named layer owning a list of weights of the model's flat parameter buffer (Keras order)."""

    def __init__(self, name, model, weight_names=(), class_name='Dense'):
        self.name = name
        self._model = model
        self.weight_names = list(weight_names)      # e.g. ['kernel', 'bias']
        self.class_name = class_name
        self.built = True

    def get_weights(self):
        P = self._model.engine.P
        return [P.p('%s/%s' % (self.name, w)).detach().cpu().numpy().copy() for w in self.weight_names]

    def set_weights(self, weights):
        P = self._model.engine.P
        if len(weights) != len(self.weight_names):
            raise ValueError("layer %s expects %d weight arrays, got %d" % (self.name, len(self.weight_names), len(weights)))
        for w, arr in zip(self.weight_names, weights):
            dst = P.p('%s/%s' % (self.name, w))
            arr = np.asarray(arr, dtype=np.float32)
            if tuple(arr.shape) != tuple(dst.shape):
                raise ValueError("layer %s weight %s: shape %s != %s" % (self.name, w, arr.shape, tuple(dst.shape)))
            dst.copy_(torch.as_tensor(arr))
        P.norms_valid = False


class OptimizerSpec:
    """What utils/weightnorm.AdamWithWeightnorm / the strings 'adam' resolve to on the HIP path."""

    def __init__(self, name='adam-wn', lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.0):
        if decay != 0.0:
            raise ValueError("learning-rate decay is not used by the reference and not supported")
        self.name, self.lr, self.beta_1, self.beta_2, self.epsilon = name, lr, beta_1, beta_2, epsilon

    @staticmethod
    def resolve(opt):
        if isinstance(opt, OptimizerSpec):
            return opt
        if opt in ('adam', 'adam-wn'):
            return OptimizerSpec(opt)
        if opt == 'rmsprop':          # Keras 2.0.0 defaults: lr 0.001, rho 0.9 (carried as beta_2), epsilon 1e-8
            return OptimizerSpec('rmsprop', lr=1e-3, beta_2=0.9, epsilon=1e-8)
        raise ValueError("optimizer %r is not supported on the HIP path (use 'adam-wn', 'adam' or 'rmsprop')" % (opt,))


def _to_dev(a, dev):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32), device=dev)


def _windows_to_dev(cur, hist, dev):
    """utils.pianoroll.Windows -> DevWindows; views of the same frame store share one device copy."""
    from .trainer import DevWindows
    stores = {}

    def one(w):
        if w is None:
            return None
        key = id(w.store)
        if key not in stores:
            stores[key] = _frames_to_dev(w.store, dev)
        return DevWindows(stores[key], torch.as_tensor(np.ascontiguousarray(w.starts, dtype=np.int64), device=dev), w.t0)
    return one(cur), one(hist)


def _frames_to_dev(a, dev):
    """Piano-roll frames for the device-resident data set: uint8 when every value is 0 or 1 (the gather converts to
    float while it assembles a batch: a quarter of the HBM footprint and of the read traffic), float32 otherwise."""
    a = np.asarray(a)
    if a.size and a.min() >= 0 and a.max() <= 1 and np.array_equal(a, a.astype(np.uint8)):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.uint8), device=dev)
    return _to_dev(a, dev)


class Model:
    """Training model; subclasses set `engine`, `layers`, `output_names`, `_split_inputs`."""

    output_names = ()
    acc_name = 'w_acc'

    def __init__(self, engine, optimizer, kl_weight, w_kl_weight, class_weight, seed=None):
        self.engine = engine
        self.optimizer = OptimizerSpec.resolve(optimizer)
        self._kl, self._wkl, self._cw = kl_weight, w_kl_weight, class_weight
        self.stop_training = False
        self.seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        self._step = None
        self._step_weights = None
        self.layers = []
        self.inputs = []
        self.history = None
        dev = engine.device
        self._acc = torch.zeros(8, dtype=torch.float32, device=dev)

    # -- structure ----------------------------------------------------------
    def get_layer(self, name):
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: " + name)

    def to_yaml(self):
        """Architecture description (written next to the weights, never read back: utils/model_utils.py:160-165)."""
        import yaml
        return yaml.safe_dump({'class_name': 'Model', 'backend': 'clvae-mi355x-hip', 'keras_version': '2.0.0',
                               'config': {'name': type(self).__name__, 'engine_config': self.engine.cfg,
                                          'batch_size': self.engine.B,
                                          'layers': [{'name': l.name, 'class_name': l.class_name,
                                                      'weights': l.weight_names} for l in self.layers]}})

    def reset_states(self):
        pass

    # -- weights ------------------------------------------------------------
    def get_weights(self):
        out = []
        for l in self.layers:
            out.extend(l.get_weights())
        return out

    def save_weights(self, filepath, overwrite=True):
        h5io.save_keras_weights(filepath, [(l.name, l.weight_names, l.get_weights()) for l in self.layers])

    def load_weights(self, filepath):
        """Keras (non by_name) semantics: layers with weights are matched BY ORDER, weights by order."""
        saved = [(n, ws) for n, ws in h5io.load_keras_weights(filepath) if len(ws) > 0]
        mine = [l for l in self.layers if l.weight_names]
        if len(saved) != len(mine):
            raise ValueError("weight file has %d layers with weights, model has %d" % (len(saved), len(mine)))
        for l, (_, ws) in zip(mine, saved):
            l.set_weights(ws)

    # -- training -------------------------------------------------------------
    def _sync_loss_weights(self):
        e = self.engine
        e.kl_weight, e.w_kl_weight, e.class_weight = get_value(self._kl), get_value(self._wkl), get_value(self._cw)
        return (e.kl_weight, e.w_kl_weight, e.class_weight)

    def _train_step(self):
        """(Re)build the captured step when the annealed loss weights changed (they are baked into the graph)."""
        w = self._sync_loss_weights()
        if self._step is None:
            rank, world = _dp()
            self._step = TrainStep(self.engine, seed=self.seed, rank=rank, world=world, optimizer=self.optimizer.name,
                                   lr=self.optimizer.lr)
            self._step_weights = w
        elif self._step_weights != w:       # same step object (streams, staging buffers): only the graphs are stale
            self._step.recapture()
            self._step_weights = w
        return self._step

    def _logs_from(self, acc, n, prefix=''):
        s = (acc.detach().cpu().numpy().astype(np.float64)) / max(n, 1)
        kl, wkl, cw = self._step_weights if self._step_weights else self._sync_loss_weights()
        names = self.output_names       # (recon, w(kl_w), w2(w_rec), z_args(kl_z))
        logs = {prefix + 'loss': s[0] + wkl * s[2] + cw * s[3] + kl * s[1],
                prefix + names[0] + '_loss': s[0], prefix + names[1] + '_loss': s[2],
                prefix + names[2] + '_loss': s[3], prefix + names[3] + '_loss': s[1],
                prefix + self.acc_name: s[4]}
        return logs

    def fit(self, x, y, shuffle=True, epochs=1, batch_size=None, callbacks=None, validation_data=None, verbose=1,
            initial_epoch=0):
        """Keras' fit() loop on device-resident data.  Under torch.distributed (one process per GPU) the
        model holds this rank's slice of every global batch: `batch_size` is the GLOBAL batch (world x the model's
        batch), rank 0 draws the epoch permutation and broadcasts it, rank r takes rows [r*B, (r+1)*B) of each
        global batch, gradients are averaged inside the step, the epoch's loss sums and the validation sums are
        all-reduced once per epoch, and callbacks that write files run on rank 0 only (Callback.rank0_only)."""
        eng = self.engine
        rank, world = _dp()
        B = eng.B
        GB = B * world
        if batch_size is not None and int(batch_size) != GB:
            raise ValueError("model was built with batch_size %d per process x %d processes, got batch_size %d"
                             % (B, world, int(batch_size)))
        cur, hist, w_true = self._split_inputs(x, y)
        tgt = self._target_of(cur, y)
        n = cur.shape[0]
        if n % GB:
            raise ValueError("number of samples %d is not a multiple of the global batch %d" % (n, GB))
        dev = eng.device
        d_cur, d_hist = self._data_to_dev(cur, hist, dev)
        d_w = _to_dev(w_true, dev)
        d_tgt = None if tgt is None else self._data_to_dev(tgt, None, dev)[0]
        val = None
        if validation_data is not None:
            vc, vh, vw = self._split_inputs(validation_data[0], validation_data[1])
            if vc.shape[0] % B:
                raise ValueError("validation samples %d not a multiple of batch_size %d" % (vc.shape[0], B))
            vt = self._target_of(vc, validation_data[1])
            if (vt is None) != (tgt is None):
                raise ValueError("training and validation targets must both be the inputs or both be separate frames")
            val = self._data_to_dev(vc, vh, dev) + (_to_dev(vw, dev), None if vt is None else self._data_to_dev(vt, None, dev)[0])
        if world > 1:           # replicas start from rank 0's weights, optimizer state and noise key
            for t in eng.P.state_tensors():     # incl. the weight-norm column state and `iterations` (resumed fits)
                dist.broadcast(t, src=0)
            # the host-side flag "vn2 describes the parameters" is per rank and did not travel: clear it everywhere, the
            # first step then takes the five-launch optimizer on EVERY rank and re-validates it (ranks must not pick
            # different Adam-WN forms, nor consume rank 0's column norms on the strength of their own flag)
            eng.P.norms_valid = False
            seed_t = torch.tensor([self.seed], dtype=torch.int64, device=dev)
            dist.broadcast(seed_t, src=0)
            if int(seed_t.item()) != self.seed:
                self.seed, self._step = int(seed_t.item()), None
        self.history = History()
        cbs = [c for c in list(callbacks or []) if rank == 0 or not getattr(c, 'rank0_only', False)] + [self.history]
        for c in cbs:
            c.set_model(self)
            c.set_params({'epochs': epochs, 'batch_size': GB, 'samples': n})
        self.stop_training = False
        for c in cbs:
            c.on_train_begin({})
        idx_dev = torch.zeros(n, dtype=torch.int64, device=dev)
        try:
            for epoch in range(initial_epoch, epochs):
                for c in cbs:
                    c.on_epoch_begin(epoch, {})
                ts = self._train_step()
                index = np.arange(n)
                if shuffle:
                    np.random.shuffle(index)                        # global np.random state, like Keras (A.4)
                idx_dev.copy_(torch.from_numpy(index))
                if world > 1:
                    dist.broadcast(idx_dev, src=0)                  # every rank walks rank 0's permutation
                self._acc.zero_()
                if ts._bound is None or ts._bound['cur'] is not d_cur:
                    # the epoch's batches are assembled by the step itself, from the device step counter: batch j of an epoch
                    # = rows idx_dev[j * GB + rank * B ..] (the permutation is rewritten in place per epoch)
                    ts.bind_batches(d_cur, d_hist, d_w, idx=idx_dev, period=n // GB, stride=GB, offset=rank * B, d_target=d_tgt)
                    ts.loss_acc = self._acc
                if world > 1 and ts.dp_trials is None and ts.ar is not None:
                    # once per fit: which weight-gradient grid is faster next to the real all-reduce on this node (timed on
                    # the first batch; parameters, optimizer state and the epoch's loss sums are restored)
                    ts.dp_trials = ts.tune_dp_schedule() or {}
                    self._acc.zero_()
                for b0 in range(0, n, GB):
                    ts.step()
                if world > 1:
                    dist.all_reduce(self._acc)                      # once per epoch (sum over ranks of per-batch means)
                logs = self._logs_from(self._acc, (n // GB) * world)
                if val is not None:
                    logs.update(self.evaluate_device(*val, prefix='val_'))
                if verbose and rank == 0:
                    print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in sorted(logs.items())))
                for c in cbs:
                    c.on_epoch_end(epoch, logs)
                if self.stop_training:
                    break
        finally:
            # the step stays bound to THIS data set only while fit() runs: a later stage_batch() / gather_batch() + step()
            # must train on what it staged (not on the bound cursor), the epoch sums must not collect other callers' steps,
            # and the device copies of the data set are released with fit()'s locals instead of living on in the step
            if self._step is not None:
                self._step.unbind_batches()
                self._step.loss_acc = None
        for c in cbs:
            c.on_train_end({})
        return self.history

    @staticmethod
    def _data_to_dev(cur, hist, dev):
        """Frames of a data set on the device: whole rows (uint8 when binary) or, for Windows views, the frame store
        plus the windows' start offsets."""
        from .utils.pianoroll import Windows
        if isinstance(cur, Windows) and (hist is None or isinstance(hist, Windows)):
            return _windows_to_dev(cur, hist, dev)
        return (_frames_to_dev(np.asarray(cur), dev), None if hist is None else _frames_to_dev(np.asarray(hist), dev))

    @staticmethod
    def _target_of(cur, y):
        """The frames the decoder output is scored against when they are NOT the input frames themselves (the scripts
        pass [recon_target, w, w, recon_target]; with --predict_next recon_target is the next frame, cl_vae/train.py:15,66
        and cl_vrnn/train.py:15,66), else None."""
        t = y[0] if isinstance(y, (list, tuple)) else None
        if t is None or t is cur:
            return None
        from .utils.pianoroll import Windows
        if isinstance(t, Windows) or isinstance(cur, Windows):
            same = isinstance(t, Windows) and isinstance(cur, Windows) and t.same_frames(cur)
        else:
            t = np.asarray(t)
            same = t.shape == np.asarray(cur).shape and (np.shares_memory(t, cur) or np.array_equal(t, cur))
        return None if same else t

    def evaluate_device(self, d_cur, d_hist, d_w, d_target=None, prefix=''):
        """Validation pass: forward + losses with the sampling noise ON (Lambda layers have no test switch,
        SURVEY.md 5.9 B10), batch-size chunks, no parameter update.  Under torch.distributed the chunks are dealt
        round-robin to the ranks and the sums are all-reduced."""
        eng = self.engine
        rank, world = _dp()
        B = eng.B
        ts = self._train_step()
        acc = torch.zeros(8, dtype=torch.float32, device=eng.device)
        n = d_cur.shape[0]
        for j, b0 in enumerate(range(0, n, B)):
            if j % world != rank:
                continue
            ts.gather_batch(d_cur, d_hist, d_w, None, row0=b0, d_target=d_target)
            ts.draw_noise(stream_offset=2 + b0 // B, row0=0)     # a validation chunk is a whole batch of its own
            eng.loss_and_grads(ts.X, ts.Xp, ts.w_true, ts.eps_w, ts.eps_z, need_grads=False, target=ts.Y)
            ops.axpy(5, 1.0, eng.scal, acc)
        if world > 1:
            dist.all_reduce(acc)
        return self._logs_from(acc, n // B, prefix)


def _dp():
    """(rank, world) of the default process group, (0, 1) without one."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def save_args_json(args_dict, path):
    with open(path, 'w') as f:
        json.dump(args_dict, f)
