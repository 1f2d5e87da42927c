"""Classifying VAE+LSTM (STORN) -- training CLI.

Same argument surface and flow as the reference's code/cl_vrnn/train.py (train :13-74, argparse
:76-118); Model.fit runs on the MI355X HIP path.

Deviation, documented: the reference sets n_classes = #unique TRAIN keys while
key indices come from a key map over all splits, and crashes in to_categorical when the train
split lacks a key.  The reference formula is used whenever it is safe; otherwise n_classes falls
back to len(key_map) with a warning.
"""
import argparse
import os
import sys

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clvae_amd  # noqa: E402,F401
from clvae_amd.cl_vrnn.model import get_model  # noqa: E402
from clvae_amd.keras_like import Variable  # noqa: E402
from clvae_amd.parallel import init_from_env  # noqa: E402
from clvae_amd.utils.model_utils import (AnnealLossWeight, get_callbacks, init_adam_wn,  # noqa: E402
                                          save_model_in_pieces, to_categorical)
from clvae_amd.utils.pianoroll import PianoData  # noqa: E402
from clvae_amd.utils.weightnorm import data_based_init  # noqa: E402


def train(args):
    # one process per GPU under torch.distributed.run: --batch_size stays the GLOBAL batch, each rank holds 1/world
    # of it (keras_like.Model.fit); a plain `python train.py` is world 1
    rank, local, world = init_from_env()
    if args.batch_size % world:
        raise SystemExit("--batch_size %d is not divisible by the %d processes" % (args.batch_size, world))
    local_batch, device = args.batch_size // world, 'cuda:%d' % local
    P = PianoData(args.train_file, batch_size=args.batch_size, seq_length=args.seq_length, step_length=1,
                  return_y_next=args.predict_next or args.use_x_prev, return_y_hist=True, squeeze_x=False,
                  squeeze_y=False, lazy=True)     # windows stay views of one uint8 frame store per split

    args.n_classes = len(np.unique(P.train_song_keys))
    max_key = max(int(P.train_song_keys.max()), int(P.valid_song_keys.max()) if len(P.valid_song_keys) else 0)
    if max_key >= args.n_classes:
        print("WARNING: key index %d >= n_classes %d (reference bug B1); using len(key_map) = %d"
              % (max_key, args.n_classes, len(P.key_map)))
        args.n_classes = len(P.key_map)
    w = to_categorical(P.train_song_keys, args.n_classes)
    wv = to_categorical(P.valid_song_keys, args.n_classes)

    print("Training with {} classes.".format(args.n_classes))
    assert not (args.predict_next and args.use_x_prev), "Can't use --predict_next if using --use_x_prev"

    callbacks = get_callbacks(args, patience=args.patience, min_epoch=max(args.kl_anneal, args.w_kl_anneal) + 1,
                              do_log=args.do_log)
    if args.kl_anneal > 0:
        assert args.kl_anneal <= args.num_epochs, "invalid kl_anneal"
        kl_weight = Variable(0.1)
        callbacks += [AnnealLossWeight(kl_weight, name="kl_weight", final_value=1.0, n_epochs=args.kl_anneal)]
    else:
        kl_weight = 1.0
    if args.w_kl_anneal > 0:
        assert args.w_kl_anneal <= args.num_epochs, "invalid w_kl_anneal"
        w_kl_weight = Variable(0.0)
        callbacks += [AnnealLossWeight(w_kl_weight, name="w_kl_weight", final_value=1.0, n_epochs=args.w_kl_anneal)]
    else:
        w_kl_weight = 1.0

    args.optimizer, was_adam_wn = init_adam_wn(args.optimizer)
    model, _ = get_model(local_batch, args.original_dim, args.intermediate_dim, args.latent_dim, args.seq_length,
                         args.n_classes, args.use_x_prev, args.optimizer, args.class_weight, kl_weight,
                         w_kl_weight=w_kl_weight, w_log_var_prior=args.w_log_var_prior,
                         seed=getattr(args, 'seed', None), device=device)
    args.optimizer = 'adam-wn' if was_adam_wn else args.optimizer
    if rank == 0:
        save_model_in_pieces(model, args)

    print((P.x_train.shape, P.y_train.shape))
    if args.use_x_prev:
        x, y = [P.y_train, P.x_train], P.y_train
        xv, yv = [P.y_valid, P.x_valid], P.y_valid
    else:
        x, y = P.x_train, P.y_train
        xv, yv = P.x_valid, P.y_valid
    ytr = [y, w, w, y]
    yva = [yv, wv, wv, yv]

    data_based_init(model, x[:100])
    history = model.fit(x, ytr, shuffle=True, epochs=args.num_epochs, batch_size=args.batch_size,
                        callbacks=callbacks, validation_data=(xv, yva))
    first = min(args.kl_anneal, args.w_kl_anneal)      # sic: min and no +1 here (reference :72)
    best_ind = np.argmin([v if i >= first else np.inf for i, v in enumerate(history.history['val_loss'])])
    best_loss = {k: history.history[k][best_ind] for k in history.history}
    return model, best_loss


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('run_name', type=str, help='tag for current run')
    parser.add_argument('--batch_size', type=int, default=200, help='batch size')
    parser.add_argument('--optimizer', type=str, default='adam-wn', help='optimizer name')
    parser.add_argument('--num_epochs', type=int, default=200, help='number of epochs')
    parser.add_argument('--original_dim', type=int, default=88, help='input dim')
    parser.add_argument('--latent_dim', type=int, default=2, help='latent dim')
    parser.add_argument('--intermediate_dim', type=int, default=88, help='intermediate dim')
    parser.add_argument('--seq_length', type=int, default=16, help='sequence length (to use as history)')
    parser.add_argument('--class_weight', type=float, default=1.0, help='relative weight on classifying key')
    parser.add_argument("--predict_next", action="store_true", help="use x_t to 'autoencode' x_{t+1}")
    parser.add_argument("--do_log", action="store_true", help="save log files")
    parser.add_argument("--w_log_var_prior", type=float, default=0.0, help="log variance prior on w")
    parser.add_argument("--kl_anneal", type=int, default=0, help="number of epochs before kl loss term is 1.0")
    parser.add_argument("--w_kl_anneal", type=int, default=0, help="number of epochs before w's kl loss term is 1.0")
    parser.add_argument('--patience', type=int, default=5, help='# of epochs, for early stopping')
    parser.add_argument("--use_x_prev", action="store_true", help="use x_{t-1} to help z_t decode x_t")
    parser.add_argument('--log_dir', type=str, default='../data/logs', help='basedir for saving log files')
    parser.add_argument('--model_dir', type=str, default='../data/models', help='basedir for saving model weights')
    parser.add_argument('--train_file', type=str, default='../data/input/JSB Chorales_Cs.pickle',
                        help='file of training data (.pickle)')
    return parser


if __name__ == '__main__':
    train(build_parser().parse_args())
