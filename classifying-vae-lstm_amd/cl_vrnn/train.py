"""Classifying VAE+LSTM -- training CLI (reference: code/cl_vrnn/train.py; flags :76-118 verbatim in
clvae_amd.cli.TABLES, flow of train() :13-74).  Model.fit runs on the MI355X HIP path."""
import os
import sys

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clvae_amd  # noqa: E402,F401
from clvae_amd.cl_vrnn.model import get_model  # noqa: E402
from clvae_amd.cli import TrainPlan, parser_for  # noqa: E402
from clvae_amd.utils.pianoroll import PianoData  # noqa: E402
from clvae_amd.utils.weightnorm import data_based_init  # noqa: E402


def count_classes(P):
    """The reference sizes the label by the distinct keys of the TRAINING songs (:27) and indexes it with key ids,
    which crashes when a key id exceeds that count: fall back to the size of the key map."""
    n = len(np.unique(P.train_song_keys))
    top = max([int(k.max()) for k in (P.train_song_keys, P.valid_song_keys) if len(k)])
    if top >= n:
        print("WARNING: key index %d >= n_classes %d (reference bug B1); using len(key_map) = %d"
              % (top, n, len(P.key_map)))
        n = len(P.key_map)
    return n


def train(args):
    plan = TrainPlan(args)
    P = PianoData(args.train_file, batch_size=args.batch_size, seq_length=args.seq_length, step_length=1,
                  return_y_next=args.predict_next or args.use_x_prev, return_y_hist=True, squeeze_x=False,
                  squeeze_y=False, lazy=True)     # windows stay views of one uint8 frame store per split
    args.n_classes = count_classes(P)
    w_train, w_valid = plan.labels(P, args.n_classes)
    print("Training with {} classes.".format(args.n_classes))
    model, _ = get_model(plan.local_batch, args.original_dim, args.intermediate_dim, args.latent_dim, args.seq_length,
                         args.n_classes, args.use_x_prev, plan.optimizer(), args.class_weight, plan.kl_weight,
                         w_kl_weight=plan.w_kl_weight, w_log_var_prior=args.w_log_var_prior,
                         seed=getattr(args, 'seed', None), device=plan.device)
    plan.describe(model)
    print((P.x_train.shape, P.y_train.shape))
    data_based_init(model, P.x_train)
    # the reference picks the best epoch from min(kl_anneal, w_kl_anneal) on here (sic, :72), not from the first epoch
    # after the ramps like cl_vae does
    return model, plan.fit(model, P, w_train, w_valid, best_from=min(args.kl_anneal, args.w_kl_anneal))


def build_parser():
    return parser_for('cl_vrnn.train')


if __name__ == '__main__':
    train(build_parser().parse_args())
