"""Callbacks and run bookkeeping of the training scripts.

Mirrors utils/model_utils.py of the reference: AnnealLossWeight (:19-50), init_adam_wn (:52-57),
EarlyStoppingAfterEpoch (:59-104), ModelCheckpointAfterEpoch (:106-140), get_callbacks
(:142-158, including the early-stop callback being appended TWICE),
save_model_in_pieces (:160-167) and the numpy helpers (:9-17).
"""
import json
import os.path

import numpy as np

from ..keras_like import Callback, get_value, set_value
from .weightnorm import AdamWithWeightnorm


def bincrossentropy(x, xhat):
    return x * np.log(np.maximum(1e-15, xhat)) + (1 - x) * np.log(np.maximum(1e-15, 1 - xhat))


def logmeanexp(vs, axis=0):
    m = np.amax(vs, axis=axis)
    return m + np.log(np.mean(np.exp(vs - m[None, :]), axis=axis))


def logsumexp(vs, axis=0):
    m = np.amax(vs, axis=axis)
    return m + np.log(np.sum(np.exp(vs - m[None, :]), axis=axis))


class AnnealLossWeight(Callback):
    """Raise a loss weight from its start value to `final_value` over `n_epochs` (linear, or sigmoid if slope>0)."""

    def __init__(self, beta, name="beta", n_epochs=10, final_value=1.0, slope=0):
        super().__init__()
        self.beta = beta
        self.name = name
        self.slope = slope
        self.n_epochs = n_epochs
        self.start_value = get_value(beta)
        self.final_value = final_value
        self.all_done = False

    def next_weight(self, x):
        if self.slope > 0:
            return 1 / (1 + np.exp(-self.slope * (x - 0.5)))
        return 1.0 * x

    def on_epoch_begin(self, epoch, logs=None):
        if self.all_done:
            return
        if epoch >= self.n_epochs:
            next_val = self.final_value
            self.all_done = True
        else:
            next_val = self.start_value + self.next_weight(1.0 * epoch / self.n_epochs) * (self.final_value - self.start_value)
        set_value(self.beta, next_val)
        print("+++++ {}: {}".format(self.name, get_value(self.beta)))


def init_adam_wn(optimizer):
    if optimizer == 'adam-wn':
        return AdamWithWeightnorm(lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-08, decay=0.0), True
    return optimizer, False


def _monitor_op(monitor, mode):
    assert mode in ['auto', 'min', 'max']
    if mode == 'min':
        return np.less
    if mode == 'max':
        return np.greater
    if 'acc' in monitor or monitor.startswith('fmeasure'):
        return np.greater
    return np.less


class EarlyStoppingAfterEpoch(Callback):
    def __init__(self, monitor='val_loss', min_epoch=0, min_delta=0, patience=0, verbose=0, mode='auto'):
        super().__init__()
        self.monitor = monitor
        self.patience = patience
        self.verbose = verbose
        self.min_epoch = min_epoch
        self.min_delta = min_delta
        self.wait = 0
        self.stopped_epoch = 0
        self.monitor_op = _monitor_op(monitor, mode)
        self.min_delta *= 1 if self.monitor_op == np.greater else -1

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.stopped_epoch = 0
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        if epoch < self.min_epoch:
            return
        current = logs.get(self.monitor)
        if self.monitor_op(current - self.min_delta, self.best):
            self.best = current
            self.wait = 0
        else:
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
            self.wait += 1


class ModelCheckpointAfterEpoch(Callback):
    rank0_only = True
    def __init__(self, filepath, monitor, min_epoch=0, save_weights_only=True, save_best_only=True, mode='auto',
                 verbose=False):
        super().__init__()
        assert save_best_only and not verbose
        self.filepath = filepath
        self.monitor = monitor
        self.min_epoch = min_epoch
        self.save_weights_only = save_weights_only
        self.monitor_op = _monitor_op(monitor, mode)
        self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_epoch_end(self, epoch, logs=None):
        if epoch < self.min_epoch:
            return
        logs = logs or {}
        filepath = self.filepath.format(epoch=epoch, **logs)
        current = logs.get(self.monitor)
        if self.monitor_op(current, self.best):
            self.best = current
            self.model.save_weights(filepath, overwrite=True)


class EpochLogger(Callback):
    rank0_only = True
    """--do_log: per-epoch scalars as JSON lines under <log_dir>/<run>/ (stands in for the TensorBoard callback)."""

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


def get_callbacks(args, patience=5, min_epoch=0, do_log=False):
    chkpt_filename = os.path.join(args.model_dir, args.run_name + '.h5')
    checkpt = ModelCheckpointAfterEpoch(chkpt_filename, min_epoch=min_epoch, monitor='val_loss',
                                        save_weights_only=True, save_best_only=True)
    callbacks = [checkpt]
    if do_log:
        callbacks.append(EpochLogger(os.path.join(args.log_dir, args.run_name)))
    if patience > 0:
        early_stop = EarlyStoppingAfterEpoch(monitor='val_loss', min_epoch=min_epoch, patience=patience, verbose=0)
        callbacks.append(early_stop)
        callbacks.append(early_stop)      # appended twice in the reference (:155,157): on_epoch_end runs twice/epoch
    return callbacks


def save_model_in_pieces(model, args):
    outfile = os.path.join(args.model_dir, args.run_name + '.yaml')
    with open(outfile, 'w') as f:
        f.write(model.to_yaml())
    outfile = os.path.join(args.model_dir, args.run_name + '.json')
    d = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else
             (v.item() if isinstance(v, np.generic) else str(v))) for k, v in vars(args).items()}
    json.dump(d, open(outfile, 'w'))


def to_categorical(y, num_classes=None):
    """keras.utils.to_categorical: zeros((len(ravel(y)), n))[arange, y] = 1."""
    y = np.array(y, dtype='int').ravel()
    if not num_classes:
        num_classes = np.max(y) + 1
    out = np.zeros((y.shape[0], num_classes))
    out[np.arange(y.shape[0]), y] = 1
    return out
