"""Classifying variational autoencoders -- training CLI (reference: code/cl_vae/train.py; flags :76-121 verbatim in
clvae_amd.cli.TABLES, flow of train() :13-74).  Model.fit runs on the MI355X HIP path."""
import os
import sys

import numpy as np

if __package__ in (None, ''):          # run as a script: make the package importable
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clvae_amd  # noqa: E402,F401
from clvae_amd.cl_vae.model import get_model  # noqa: E402
from clvae_amd.cli import BF16_FLAGS, TrainPlan, parser_for  # noqa: E402
from clvae_amd.utils.pianoroll import PianoData  # noqa: E402
from clvae_amd.utils.weightnorm import data_based_init  # noqa: E402

SPLITS = ('x_train', 'x_valid', 'x_test', 'y_train', 'y_valid', 'y_test')


def flatten_windows(P, args):
    """--seq_length > 1 (cl_vae/train.py:18-27): a sample is the window's frames side by side, restricted to the
    notes that sound anywhere in the data set; original_dim becomes (#such notes) * seq_length."""
    sounding = np.vstack([getattr(P, nm) for nm in SPLITS]).sum(axis=(0, 1)) > 0
    for nm in SPLITS:
        a = getattr(P, nm)
        setattr(P, nm, a[:, :, sounding].reshape((len(a), -1)))
    args.original_dim = int(sounding.sum() * args.seq_length)


def train(args):
    plan = TrainPlan(args)
    P = PianoData(args.train_file, batch_size=args.batch_size, seq_length=args.seq_length, step_length=1,
                  return_y_next=args.predict_next or args.use_x_prev, squeeze_x=True, squeeze_y=True)
    if args.seq_length > 1:
        flatten_windows(P, args)
    args.n_classes = len(np.unique(P.train_song_keys))
    w_train, w_valid = plan.labels(P, args.n_classes)
    model, enc_model = get_model(plan.local_batch, args.original_dim, (args.intermediate_dim, args.latent_dim),
                                 (args.intermediate_class_dim, args.n_classes), plan.optimizer(), args.class_weight,
                                 plan.kl_weight, use_x_prev=args.use_x_prev, w_kl_weight=plan.w_kl_weight,
                                 w_log_var_prior=args.w_log_var_prior, seed=getattr(args, 'seed', None),
                                 device=plan.device, bf16=getattr(args, 'bf16', False))
    plan.describe(model)
    data_based_init(model, P.x_train[:100])
    return model, plan.fit(model, P, w_train, w_valid)


def build_parser():
    return parser_for('cl_vae.train')


if __name__ == '__main__':
    train(parser_for('cl_vae.train', BF16_FLAGS).parse_args())
