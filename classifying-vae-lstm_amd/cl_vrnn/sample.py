"""cl_vrnn sampling CLI (reference: code/cl_vrnn/sample.py; flags :49-72 verbatim in clvae_amd.cli.TABLES)."""
import os
import sys

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clvae_amd  # noqa: E402,F401
from clvae_amd.cl_vrnn import model as M  # noqa: E402
from clvae_amd.cli import DEVICE_LOOP_FLAGS, parser_for  # noqa: E402
from clvae_amd.utils.midi_utils import write_sample  # noqa: E402
from clvae_amd.utils.model_utils import to_categorical  # noqa: E402
from clvae_amd.utils.pianoroll import PianoData  # noqa: E402


def seed_windows(P, key, n):
    """Up to n test windows in random order (one np.random.shuffle), all keys or only those in `key`."""
    name_of = {idx: name for name, idx in P.key_map.items()}
    picks = np.arange(len(P.test_song_keys))
    if key is not None:
        picks = picks[np.array([name_of[k] for k in P.test_song_keys]) == key]
    np.random.shuffle(picks)
    return picks[:n]


def gen_samples(P, dec_model, w_enc_model, z_enc_model, args, margs, model=None):
    """Seed windows -> generated continuations; writes <run>_<j>.mid and the seed as <run><j>_seed_<i>.mid.
    With `model` the frame loops of all seeds run together on the device (Philox noise)."""
    half_speed = 'jsb' in args.train_file.lower()
    picks = seed_windows(P, args.c, args.n)
    label_of = lambda i: None if args.infer_w else to_categorical(P.test_song_keys[i], margs['n_classes'])
    if model is not None and len(picks):
        ws = [label_of(i) for i in picks]
        if args.infer_w:
            ws = [M.infer_label(w_enc_model, P.x_test[i], margs['seq_length'], discrete=args.discrete_w) for i in picks]
        rolls = list(M.generate_samples_device(model, np.stack([P.x_test[i] for i in picks]), args.t, np.vstack(ws),
                                               seed=getattr(args, 'seed', 0)))
    else:
        rolls = [M.generate_sample(dec_model, w_enc_model, z_enc_model, P.x_test[i], args.t, margs['use_x_prev'],
                                   w_val=label_of(i), w_discrete=args.discrete_w, seq_length=margs['seq_length'])
                 for i in picks]
    for j, (i, roll) in enumerate(zip(picks, rolls)):
        write_sample(roll, args.sample_dir, '%s_%d' % (args.run_name, j), half_speed)
        write_sample(P.x_test[i], args.sample_dir, '%s%d_seed_%d' % (args.run_name, j, i), half_speed)
    return rolls


def sample(args):
    model, _, margs = M.load_model(args.model_file, optimizer='adam')
    dims = (margs['intermediate_dim'], margs['latent_dim'])
    w_enc = M.make_w_encoder(model, margs['original_dim'], margs['n_classes'], margs['seq_length'])
    z_enc = M.make_z_encoder(model, margs['original_dim'], margs['n_classes'], dims)
    dec = M.make_decoder(model, margs['original_dim'], margs['intermediate_dim'], margs['latent_dim'],
                         margs['n_classes'], margs['use_x_prev'])
    P = PianoData(args.train_file, batch_size=1, seq_length=args.t, squeeze_x=False)
    # the reference's host loop (np.random) for every -n; --device_loop opts into the device-side loop (Philox noise)
    on_device = bool(getattr(args, 'device_loop', False)) and not getattr(args, 'host_loop', False)
    return gen_samples(P, dec, w_enc, z_enc, args, margs, model=model if on_device else None)


def build_parser():
    return parser_for('cl_vrnn.sample')


if __name__ == '__main__':
    sample(parser_for('cl_vrnn.sample', DEVICE_LOOP_FLAGS).parse_args())
