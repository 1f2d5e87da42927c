"""cl_vae sampling CLI (reference: code/cl_vae/sample.py; flags :35-61 verbatim in clvae_amd.cli.TABLES)."""
import os
import sys

import numpy as np

if __package__ in (None, ''):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clvae_amd  # noqa: E402,F401
from clvae_amd.cl_vae import model as M  # noqa: E402
from clvae_amd.cli import DEVICE_LOOP_FLAGS, parser_for  # noqa: E402
from clvae_amd.utils.midi_utils import write_sample  # noqa: E402
from clvae_amd.utils.model_utils import to_categorical  # noqa: E402
from clvae_amd.utils.pianoroll import PianoData  # noqa: E402


class Sampler:
    """The trained model's three inference views plus the test split the seed frames come from."""

    def __init__(self, args):
        self.args = args
        self.model, _, self.margs = M.load_model(args.model_file, no_x_prev=args.no_x_prev,
                                                 batch_size=max(1, args.n if on_device(args) else 1))
        m, dims = self.margs, (self.margs['intermediate_dim'], self.margs['latent_dim'])
        self.w_enc = M.make_w_encoder(self.model, m['original_dim'])
        self.z_enc = M.make_z_encoder(self.model, m['original_dim'], m['n_classes'], dims)
        self.dec = M.make_decoder(self.model, dims, m['n_classes'], use_x_prev=m['use_x_prev'])
        self.data = PianoData(args.train_file, batch_size=1, seq_length=args.t, squeeze_x=True)

    def pick_seed(self):
        """(first frame of a random test window, its key's one-hot or None with --infer_w); one np.random draw"""
        i = np.random.choice(range(len(self.data.x_test)))
        w = None if self.args.infer_w else to_categorical(self.data.test_song_keys[i], self.margs['n_classes'])
        return self.data.x_test[i][0], w

    def one(self, name):
        x_seed, w_val = self.pick_seed()
        roll = M.generate_sample(self.dec, self.w_enc, self.z_enc, x_seed, self.args.t, w_val=w_val,
                                 use_z_prior=self.args.use_z_prior, use_x_prev=self.margs['use_x_prev'])
        write_sample(roll, self.args.sample_dir, name, True)
        return roll

    def many_on_device(self, names):
        """All samples in one device-side frame loop (Philox noise; the seeds are drawn like one() draws them)."""
        seeds, ws = zip(*[self.pick_seed() for _ in names])
        if self.args.infer_w:
            ws = [M.sample_w(self.w_enc.predict(s[None, :]), add_noise=False) for s in seeds]
        rolls = M.generate_samples_device(self.model, np.stack(seeds), self.args.t, np.vstack(ws),
                                          seed=getattr(self.args, 'seed', 0), use_z_prior=self.args.use_z_prior)
        for roll, name in zip(rolls, names):
            write_sample(roll, self.args.sample_dir, name, True)
        return list(rolls)


def make_sample(P, dec_model, w_enc_model, z_enc_model, args, margs):
    """One sample from explicit sub-models (the reference's helper, :8-19)."""
    i = np.random.choice(range(len(P.x_test)))
    w_val = None if args.infer_w else to_categorical(P.test_song_keys[i], margs['n_classes'])
    roll = M.generate_sample(dec_model, w_enc_model, z_enc_model, P.x_test[i][0], args.t, w_val=w_val,
                             use_z_prior=args.use_z_prior, use_x_prev=margs['use_x_prev'])
    write_sample(roll, args.sample_dir, args.run_name, True)
    return roll


def on_device(args):
    """Where the frame loop runs: like the reference (host loop, np.random) for every -n unless --device_loop asks for
    the device-side loop (Philox noise: other samples for the same np.random.seed, so it is opt-in)."""
    return bool(getattr(args, 'device_loop', False)) and not getattr(args, 'host_loop', False)


def sample(args):
    s = Sampler(args)
    names = ['%s_%d' % (args.run_name, i) for i in range(args.n)]
    return s.many_on_device(names) if on_device(args) else [s.one(nm) for nm in names]


def build_parser():
    return parser_for('cl_vae.sample')


if __name__ == '__main__':
    sample(parser_for('cl_vae.sample', DEVICE_LOOP_FLAGS).parse_args())
