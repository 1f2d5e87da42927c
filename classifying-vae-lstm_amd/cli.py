"""The four command-line tools' argument tables and the parts of a run that do not depend on the model family.

The reference spells the same flags out four times (cl_vae/train.py:76-121, cl_vrnn/train.py:76-118,
cl_vae/sample.py:35-61, cl_vrnn/sample.py:49-72).  Here each tool is a list of `Flag` rows (names, type, default, help;
verbatim, they are the interface) and the two training scripts share `TrainPlan`: process layout under
torch.distributed, loss-weight ramps, callbacks, the `[y, w, w, y]` target wiring, fit, and the pick of the best epoch.
"""
import argparse
from collections import namedtuple

import numpy as np

from .keras_like import Variable
from .parallel import init_from_env
from .utils.model_utils import (AnnealLossWeight, best_epoch, get_callbacks, init_adam_wn, save_model_in_pieces,
                                to_categorical)

Flag = namedtuple('Flag', 'names kind default help')
ON = 'store_true'          # kind of a switch


def _train_flags(batch_size, seq_length, seq_help, extra=()):
    return [
        Flag(('run_name',), str, None, 'tag for current run'),
        Flag(('--batch_size',), int, batch_size, 'batch size'),
        Flag(('--optimizer',), str, 'adam-wn', 'optimizer name'),
        Flag(('--num_epochs',), int, 200, 'number of epochs'),
        Flag(('--original_dim',), int, 88, 'input dim'),
        Flag(('--intermediate_dim',), int, 88, 'intermediate dim'),
        Flag(('--latent_dim',), int, 2, 'latent dim'),
        Flag(('--seq_length',), int, seq_length, seq_help),
        Flag(('--class_weight',), float, 1.0, 'relative weight on classifying key'),
        Flag(('--w_log_var_prior',), float, 0.0, 'w log var prior'),
        *extra,
        Flag(('--do_log',), ON, False, 'save log files'),
        Flag(('--predict_next',), ON, False, "use x_t to 'autoencode' x_{t+1}"),
        Flag(('--use_x_prev',), ON, False, 'use x_{t-1} to help z_t decode x_t'),
        Flag(('--patience',), int, 5, '# of epochs, for early stopping'),
        Flag(('--kl_anneal',), int, 0, 'number of epochs before kl loss term is 1.0'),
        Flag(('--w_kl_anneal',), int, 0, "number of epochs before w's kl loss term is 1.0"),
        Flag(('--log_dir',), str, '../data/logs', 'basedir for saving log files'),
        Flag(('--model_dir',), str, '../data/models', 'basedir for saving model weights'),
        Flag(('--train_file',), str, '../data/input/JSB Chorales_Cs.pickle', 'file of training data (.pickle)'),
    ]


_SAMPLE_TAIL = [
    Flag(('--sample_dir',), str, '../data/samples', 'basedir for saving output midi files'),
]
_SAMPLE_FILES = [
    Flag(('-i', '--model_file'), str, '', 'preload model weights (no training)'),
    Flag(('--train_file',), str, '../data/input/JSB Chorales_Cs.pickle', 'file of training data (.pickle)'),
]

TABLES = {
    'cl_vae.train': _train_flags(100, 1, 'sequence length (concat)',
                                 extra=[Flag(('--intermediate_class_dim',), int, 88, 'intermediate dims for classes')]),
    'cl_vrnn.train': _train_flags(200, 16, 'sequence length'),
    'cl_vae.sample': [
        Flag(('run_name',), str, None, 'tag for current run'),
        Flag(('-n',), int, 1, 'number of samples'),
        Flag(('--use_z_prior',), ON, False, 'sample z from standard normal at each timestep'),
        Flag(('-t',), int, 32, 'number of timesteps per sample'),
        Flag(('--infer_w',), ON, False, 'infer w when generating'),
        Flag(('--no_x_prev',), ON, False, 'override use_x_prev'),
        *_SAMPLE_TAIL,
        Flag(('--model_dir',), str, '../data/models', 'basedir for saving model weights'),
        *_SAMPLE_FILES,
    ],
    'cl_vrnn.sample': [
        Flag(('run_name',), str, None, 'tag for current run'),
        Flag(('--infer_w',), ON, False, 'infer w when generating'),
        Flag(('--discrete_w',), ON, False, 'sample discrete w when generating'),
        Flag(('-t',), int, 32, 'number of timesteps per sample'),
        Flag(('-n',), int, 1, 'number of samples'),
        Flag(('-c',), str, None, 'set key of seed sample'),
        *_SAMPLE_TAIL,
        *_SAMPLE_FILES,
    ],
}

# switches of THIS implementation (not in the reference): where the frame loop of sample.py runs
# cl_vae/train.py only, next to the reference's flags
BF16_FLAGS = [
    Flag(('--bf16',), ON, False, 'Dense products of the fused training step on the bf16 matrix cores (fp32 accumulate)'),
]

DEVICE_LOOP_FLAGS = [
    Flag(('--device_loop',), ON, False, 'generate all -n samples in one device-side frame loop (Philox noise; opt-in: the default is the reference host loop)'),
    Flag(('--host_loop',), ON, False, 'frame loop on the host with np.random, like the reference (the default; overrides --device_loop)'),
    Flag(('--seed',), int, 0, 'noise key of the device-side loop'),
]


def parser_for(tool, extra=()):
    p = argparse.ArgumentParser()
    for f in list(TABLES[tool]) + list(extra):
        if f.kind == ON:
            p.add_argument(*f.names, action=ON, help=f.help)
        elif f.names[0].startswith('-'):
            p.add_argument(*f.names, type=f.kind, default=f.default, help=f.help)
        else:
            p.add_argument(*f.names, type=f.kind, help=f.help)
    return p


class TrainPlan:
    """Everything of train() that is the same for cl_vae and cl_vrnn.

    Under torch.distributed.run (one process per GPU) `--batch_size` stays the GLOBAL batch: the model is built for
    batch_size / world rows and Model.fit shards every batch (keras_like.Model.fit); a plain `python train.py` is
    world 1."""

    def __init__(self, args):
        self.args = args
        self.rank, local, self.world = init_from_env()
        if args.batch_size % self.world:
            raise SystemExit("--batch_size %d is not divisible by the %d processes" % (args.batch_size, self.world))
        self.local_batch = args.batch_size // self.world
        self.device = 'cuda:%d' % local
        if args.predict_next and args.use_x_prev:
            raise AssertionError("Can't use --predict_next if using --use_x_prev")
        # epochs before this one are never checkpointed, never stop the run and never count as "best"
        self.first_epoch = max(args.kl_anneal, args.w_kl_anneal) + 1
        self.callbacks = get_callbacks(args, patience=args.patience, min_epoch=self.first_epoch, do_log=args.do_log)
        self.kl_weight = self._ramped('kl_weight', args.kl_anneal, start=0.1)
        self.w_kl_weight = self._ramped('w_kl_weight', args.w_kl_anneal, start=0.0)

    def _ramped(self, name, n_epochs, start):
        """1.0, or a variable that a callback raises from `start` to 1.0 over the first n_epochs epochs"""
        if n_epochs <= 0:
            return 1.0
        if n_epochs > self.args.num_epochs:
            raise AssertionError("invalid " + name.replace('_weight', '_anneal'))
        var = Variable(start)
        self.callbacks.append(AnnealLossWeight(var, name=name, final_value=1.0, n_epochs=n_epochs))
        return var

    def labels(self, P, n_classes):
        return to_categorical(P.train_song_keys, n_classes), to_categorical(P.valid_song_keys, n_classes)

    def optimizer(self):
        """The optimizer object for get_model; `args.optimizer` keeps the flag's string for the run's JSON."""
        opt, _ = init_adam_wn(self.args.optimizer)
        return opt

    def describe(self, model):
        if self.rank == 0:
            save_model_in_pieces(model, self.args)

    def fit(self, model, P, w_train, w_valid, best_from=None):
        """inputs [y, x] (current frames, history) with --use_x_prev, else x; targets [recon, w, w, recon].
        Returns the history entries of the best epoch (lowest val_loss from epoch `best_from` on; default: the first
        epoch after the ramps)."""
        a = self.args
        if a.use_x_prev:
            x_tr, x_va = [P.y_train, P.x_train], [P.y_valid, P.x_valid]
        else:
            x_tr, x_va = P.x_train, P.x_valid
        hist = model.fit(x_tr, [P.y_train, w_train, w_train, P.y_train], shuffle=True, epochs=a.num_epochs,
                         batch_size=a.batch_size, callbacks=self.callbacks,
                         validation_data=(x_va, [P.y_valid, w_valid, w_valid, P.y_valid]))
        at = best_epoch(hist.history['val_loss'], self.first_epoch if best_from is None else best_from)
        return {k: v[at] for k, v in hist.history.items()}
