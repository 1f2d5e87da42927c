"""Run bookkeeping of the training scripts: loss-weight ramps, best-value tracking, checkpoints, early stopping.

Same names, constructor arguments and observable behaviour as the reference's utils/model_utils.py (AnnealLossWeight
:19-50, init_adam_wn :52-57, EarlyStoppingAfterEpoch :59-104, ModelCheckpointAfterEpoch :106-140, get_callbacks
:142-158 with the early-stop callback listed TWICE -- SURVEY.md 5.9 B4 --, save_model_in_pieces :160-167, numpy helpers
:9-17), built here from two pieces:

  * `Ramp`       a loss weight as a function of the epoch (linear, or a logistic curve when slope > 0);
  * `BestSoFar`  "is this epoch's monitored value better than every earlier one?", shared by the checkpoint callback,
                 the early-stop callback and the scripts' pick of the best epoch (`best_epoch`).
"""
import json
import os.path

import numpy as np

from ..keras_like import Callback, get_value, set_value
from .weightnorm import AdamWithWeightnorm


# ---- numpy helpers ----------------------------------------------------------
def bincrossentropy(x, xhat):
    """log-likelihood of binary x under Bernoulli(xhat), elementwise, with the probabilities floored at 1e-15"""
    floor = lambda p: np.log(np.maximum(1e-15, p))
    return x * floor(xhat) + (1 - x) * floor(1 - xhat)


def _log_reduce_exp(vs, axis, reduce):
    peak = np.amax(vs, axis=axis)
    return peak + np.log(reduce(np.exp(vs - peak[None, :]), axis=axis))


def logmeanexp(vs, axis=0):
    return _log_reduce_exp(vs, axis, np.mean)


def logsumexp(vs, axis=0):
    return _log_reduce_exp(vs, axis, np.sum)


def to_categorical(y, num_classes=None):
    """keras.utils.to_categorical: one row per entry of ravel(y); a scalar gives shape (1, n)."""
    idx = np.array(y, dtype='int').ravel()
    n = int(num_classes) if num_classes else int(idx.max()) + 1
    return (idx[:, None] == np.arange(n)[None, :]).astype(np.float64)


# ---- loss-weight ramps --------------------------------------------------------
class Ramp:
    """start -> final over n_epochs: value(epoch) = start + shape(epoch / n_epochs) * (final - start), then final."""

    def __init__(self, start, final, n_epochs, slope=0):
        self.start, self.final, self.n_epochs, self.slope = start, final, n_epochs, slope

    def shape(self, x):
        return 1 / (1 + np.exp(-self.slope * (x - 0.5))) if self.slope > 0 else 1.0 * x

    def finished(self, epoch):
        return epoch >= self.n_epochs

    def value(self, epoch):
        if self.finished(epoch):
            return self.final
        return self.start + self.shape(1.0 * epoch / self.n_epochs) * (self.final - self.start)


class AnnealLossWeight(Callback):
    """Sets the loss-weight variable `beta` at the start of every epoch until its ramp is over."""

    def __init__(self, beta, name="beta", n_epochs=10, final_value=1.0, slope=0):
        super().__init__()
        self.beta, self.name = beta, name
        self.ramp = Ramp(get_value(beta), final_value, n_epochs, slope)
        self.all_done = False

    # the reference's attribute / method names, kept for scripts that poke at them
    n_epochs = property(lambda self: self.ramp.n_epochs)
    start_value = property(lambda self: self.ramp.start)
    final_value = property(lambda self: self.ramp.final)
    slope = property(lambda self: self.ramp.slope)

    def next_weight(self, x):
        return self.ramp.shape(x)

    def on_epoch_begin(self, epoch, logs=None):
        if self.all_done:
            return
        self.all_done = self.ramp.finished(epoch)
        set_value(self.beta, self.ramp.value(epoch))
        print("+++++ {}: {}".format(self.name, get_value(self.beta)))


# ---- best-value tracking ------------------------------------------------------
def _lower_is_better(monitor, mode):
    if mode not in ('auto', 'min', 'max'):
        raise AssertionError("mode must be auto, min or max")
    if mode != 'auto':
        return mode == 'min'
    return not ('acc' in monitor or monitor.startswith('fmeasure'))


class BestSoFar:
    """Tracks the best value of one monitored quantity; `margin` must be beaten for a value to count as better."""

    def __init__(self, monitor, mode='auto', margin=0):
        self.monitor = monitor
        self.lower = _lower_is_better(monitor, mode)
        self.margin = abs(margin)
        self.reset()

    def reset(self):
        self.best = np.inf if self.lower else -np.inf

    def offer(self, value):
        """True (and remembered) when `value` beats the best so far by more than the margin."""
        better = value + self.margin < self.best if self.lower else value - self.margin > self.best
        if better:
            self.best = value
        return bool(better)


def best_epoch(values, first_epoch=0, lower=True):
    """Index of the best entry among epochs >= first_epoch (ties: the earliest), as the train scripts pick it."""
    v = np.asarray(values, dtype=np.float64)
    masked = np.where(np.arange(len(v)) >= first_epoch, v if lower else -v, np.inf)
    return int(np.argmin(masked))


class _AfterEpoch(Callback):
    """A callback that ignores the epochs before `min_epoch` and reads one monitored value per epoch."""

    def __init__(self, monitor, min_epoch, mode, margin=0):
        super().__init__()
        self.monitor, self.min_epoch = monitor, min_epoch
        self.tracker = BestSoFar(monitor, mode, margin)

    best = property(lambda self: self.tracker.best)
    monitor_op = property(lambda self: np.less if self.tracker.lower else np.greater)

    def on_epoch_end(self, epoch, logs=None):
        if epoch >= self.min_epoch:
            self.after_epoch(epoch, logs or {}, (logs or {}).get(self.monitor))


class EarlyStoppingAfterEpoch(_AfterEpoch):
    """Stops training once `patience` calls in a row (after min_epoch) saw no improvement.  The counter moves once per
    CALL: get_callbacks lists the object twice, so an epoch without improvement counts twice."""

    def __init__(self, monitor='val_loss', min_epoch=0, min_delta=0, patience=0, verbose=0, mode='auto'):
        super().__init__(monitor, min_epoch, mode, margin=min_delta)
        self.patience, self.verbose, self.min_delta = patience, verbose, min_delta
        self.wait = self.stopped_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = self.stopped_epoch = 0
        self.tracker.reset()

    def after_epoch(self, epoch, logs, current):
        if self.tracker.offer(current):
            self.wait = 0
            return
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            self.model.stop_training = True
        self.wait += 1


class ModelCheckpointAfterEpoch(_AfterEpoch):
    """Saves the weights whenever the monitored value is the best so far (after min_epoch)."""
    rank0_only = True

    def __init__(self, filepath, monitor, min_epoch=0, save_weights_only=True, save_best_only=True, mode='auto',
                 verbose=False):
        if not save_best_only or verbose:
            raise AssertionError("only the reference's configuration (best only, silent) exists")
        super().__init__(monitor, min_epoch, mode)
        self.filepath, self.save_weights_only = filepath, save_weights_only

    def after_epoch(self, epoch, logs, current):
        if self.tracker.offer(current):
            self.model.save_weights(self.filepath.format(epoch=epoch, **logs), overwrite=True)


class EpochLogger(Callback):
    """--do_log: per-epoch scalars as JSON lines under <log_dir>/<run>/ (stands in for the TensorBoard callback)."""
    rank0_only = True

    def __init__(self, log_dir):
        super().__init__()
        self.log_dir = log_dir

    def on_train_begin(self, logs=None):
        os.makedirs(self.log_dir, exist_ok=True)
        self.f = open(os.path.join(self.log_dir, 'epochs.jsonl'), 'a')

    def on_epoch_end(self, epoch, logs=None):
        self.f.write(json.dumps(dict(epoch=epoch, **{k: float(v) for k, v in (logs or {}).items()})) + "\n")
        self.f.flush()

    def on_train_end(self, logs=None):
        self.f.close()


# ---- what the scripts call ------------------------------------------------------
def init_adam_wn(optimizer):
    """'adam-wn' -> (AdamWithWeightnorm with the reference's hyper-parameters, True); anything else passes through."""
    if optimizer != 'adam-wn':
        return optimizer, False
    return AdamWithWeightnorm(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-08, decay=0.0), True


def get_callbacks(args, patience=5, min_epoch=0, do_log=False):
    """[checkpoint, (logger), (early stop, early stop)] on val_loss, all silent before min_epoch."""
    weights_file = os.path.join(args.model_dir, args.run_name + '.h5')
    out = [ModelCheckpointAfterEpoch(weights_file, monitor='val_loss', min_epoch=min_epoch, save_weights_only=True,
                                     save_best_only=True)]
    if do_log:
        out.append(EpochLogger(os.path.join(args.log_dir, args.run_name)))
    if patience > 0:
        out += 2 * [EarlyStoppingAfterEpoch(monitor='val_loss', min_epoch=min_epoch, patience=patience, verbose=0)]
    return out


def _plain(v):
    if isinstance(v, (int, float, str, bool, type(None))):
        return v
    return v.item() if isinstance(v, np.generic) else str(v)


def save_model_in_pieces(model, args):
    """<model_dir>/<run>.yaml (architecture text) and <run>.json (the run's arguments: what load_model rebuilds from)."""
    stem = os.path.join(args.model_dir, args.run_name)
    with open(stem + '.yaml', 'w') as f:
        f.write(model.to_yaml())
    with open(stem + '.json', 'w') as f:
        json.dump({k: _plain(v) for k, v in vars(args).items()}, f)
