"""Classifying VAE+LSTM (cl_vrnn, STORN-like) on the MI355X HIP path.

Same module surface as the reference's code/cl_vrnn/model.py: get_model (:164-267), load_model
(:269-282), make_w_encoder (:98-114), make_z_encoder (:116-136), make_decoder (:138-162),
generate_sample (:9-60), sample_x / sample_w_discrete / sample_w / sample_z (:62-96).
`latent_dims` of make_z_encoder is the 2-tuple (latent_dim_0, latent_dim) of the reference's
tuple parameter.

Deviation, documented: the reference's make_z_encoder builds a FRESH,
randomly initialised encoder LSTM and copies only the Z heads; here the trained encoder LSTM is
used.  `emulate_fresh_encoder=True` reproduces the reference behaviour.
"""
import json

import numpy as np
import torch

from .. import sampling
from ..engine import VrnnEngine, vrnn_param_shapes
from ..initializers import glorot_uniform, init_weights, orthogonal
from ..keras_like import Layer, Model, get_value
from ..utils.pianoroll import Windows


# --------------------------------------------------------------------------- #
# host-side sampling (clvae_amd.sampling holds the shared pieces; these are cl_vrnn's bindings)
# --------------------------------------------------------------------------- #
def sample_x(x_mean):
    """x ~ Bernoulli(x_mean) as 0./1. (reference :62-63: uniforms shaped like the squeezed mean)"""
    return sampling.draw_frame(x_mean, flat=False)


def sample_w_discrete(w):
    """A one-hot label drawn from the label vector w (reference :65-68)."""
    return sampling.one_hot_draw(w)


def sample_w(args, nsamps=1, nrm_samp=False, add_noise=True):
    """Label sample from (w_mean, w_log_var) (reference :70-88)."""
    w_mean, w_log_var = args
    return sampling.logistic_normal(w_mean, w_log_var, nsamps, normal_only=nrm_samp, add_noise=add_noise)


def sample_z(args, nsamps=1):
    """Latent sample from (Z_mean, Z_log_var) (reference :90-96)."""
    return sampling.gaussian(args[0], args[1], nsamps)


def label_windows(x_seed, seq_length):
    """The windows the label is inferred from: consecutive chunks of seq_length frames; an incomplete chunk is
    skipped.  The reference walks the chunk starts up to x_seed.shape[1] -- the FEATURE dimension (:35, SURVEY.md 5.9
    B5) --, so a seed shorter than 88 frames is covered entirely and the starts beyond its end give empty chunks."""
    starts = range(0, x_seed.shape[1], seq_length)
    return [x_seed[i:i + seq_length][None, :] for i in starts if len(x_seed[i:i + seq_length]) == seq_length]


def infer_label(w_enc_model, x_seed, seq_length, add_noise=False, discrete=False):
    """w [1,C]: the mean of the per-window label samples, optionally replaced by a one-hot draw from it."""
    wins = label_windows(x_seed, seq_length)
    if not wins:
        raise ValueError("need at least one window of seq_length frames to infer w from")
    w = np.vstack([sample_w(w_enc_model.predict(win), add_noise=add_noise) for win in wins]).mean(axis=0)[None, :]
    return sample_w_discrete(w[0])[None, :] if discrete else w


def generate_sample(dec_model, w_enc_model, z_enc_model, x_seed, nsteps, use_x_prev, w_val=None, do_reset=True,
                    seq_length=None, w_sample=False, w_discrete=False):
    """Continue a seed (reference :9-60): the stateful step models are teacher-forced through the seed frames
    [S,D] (S = 0 for a single seed frame [D], which is then the first input), then run `nsteps` frames on their own
    output; returns the free-running frames only."""
    loop = sampling.HostFrameLoop(dec_model, w_enc_model, z_enc_model, sample_x, sample_w, sample_z)
    if do_reset:
        loop.reset()
    n_seed = len(x_seed) if x_seed.ndim > 1 else 0
    w = w_val if w_val is not None else infer_label(w_enc_model, x_seed, seq_length, w_sample, w_discrete)
    x_in = None if n_seed else x_seed[None, None, :]
    frames = np.zeros((n_seed + nsteps, x_seed.shape[-1]))
    for t in range(n_seed + nsteps):
        if t < n_seed:
            x_in = x_seed[t][None, None, :]
        z = loop.latent(x_in, w)
        frames[t] = x_in = loop.frame([z, x_in, w] if use_x_prev else [z, w])
    return frames[n_seed:]


# --------------------------------------------------------------------------- #
# models
# --------------------------------------------------------------------------- #
def _dev(a, dev, shape):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32), device=dev).reshape(shape)


class ClVrnnModel(Model):
    output_names = ('X_decoded_mean', 'W', 'W2', 'Z_args')
    acc_name = 'W_acc'

    def __init__(self, engine, optimizer, kl_weight, w_kl_weight, class_weight, use_x_prev, seed=None):
        super().__init__(engine, optimizer, kl_weight, w_kl_weight, class_weight, seed)
        self.use_x_prev = use_x_prev
        self.inputs = ['current', 'history'] if use_x_prev else ['current']
        L = lambda n, w=(), c='Dense': Layer(n, self, w, c)
        kb, lstm = ('kernel', 'bias'), ('kernel', 'recurrent_kernel', 'bias')
        order = [L('current', c='InputLayer'), L('flatten_1', c='Flatten'), L('hW', kb), L('Wargs', kb),
                 L('lambda_1', c='Lambda'), L('lambda_2', c='Lambda'), L('W', c='Lambda'),
                 L('repeat_vector_1', c='RepeatVector'), L('concatenate_1', c='Concatenate'),
                 L('encoder_h', lstm, 'LSTM'), L('Z_mean', kb, 'TimeDistributed'),
                 L('Z_log_var', kb, 'TimeDistributed')]
        if use_x_prev:
            order += [L('history', c='InputLayer')]
        order += [L('lambda_3', c='Lambda')]
        if use_x_prev:
            order += [L('concatenate_2', c='Concatenate')]
        order += [L('repeat_vector_2', c='RepeatVector'),
                  L('concatenate_3' if use_x_prev else 'concatenate_2', c='Concatenate'),
                  L('decoder_h', lstm, 'LSTM'), L('X_decoded_mean', kb, 'TimeDistributed'), L('W2', c='Lambda'),
                  L('Z_args', c='Concatenate')]
        self.layers = order

    def _split_inputs(self, x, y):
        if self.use_x_prev:
            cur, hist = x[0], x[1]
        else:
            cur, hist = (x[0] if isinstance(x, (list, tuple)) else x), None
        keep = lambda a: a if isinstance(a, Windows) else np.asarray(a)       # lazy windows stay views (8f4)
        return keep(cur), (None if hist is None else keep(hist)), np.asarray(y[1])

    def predict(self, x, batch_size=None, verbose=0):
        """[X_decoded_mean, W, W2, Z_args] with freshly drawn noise, in chunks of the model's batch size."""
        eng = self.engine
        B, T, D, dev = eng.B, eng.cfg['T'], eng.cfg['D'], eng.device
        cur = np.asarray(x[0] if self.use_x_prev else (x[0] if isinstance(x, (list, tuple)) else x))
        hist = np.asarray(x[1]) if self.use_x_prev else None
        n = cur.shape[0]
        if n % B:
            raise ValueError("predict needs a multiple of the fixed batch size %d" % B)
        ts = self._train_step()
        outs = [[], [], [], []]
        for b0 in range(0, n, B):
            ts.X.copy_(_dev(cur[b0:b0 + B], dev, (B, T, D)))
            if hist is not None:
                ts.Xp.copy_(_dev(hist[b0:b0 + B], dev, (B, T, D)))
            ts.draw_noise(stream_offset=1000 + b0 // B)
            eng.forward(ts.X, ts.Xp, ts.eps_w, ts.eps_z)
            outs[0].append(eng.x_hat().view(B, T, D).cpu().numpy())
            w = eng.W.cpu().numpy()
            outs[1].append(w); outs[2].append(w + 1e-10)
            outs[3].append(eng.zargs.view(B, T, -1).cpu().numpy())
        return [np.concatenate(o) for o in outs]


class _Stateful:
    def __init__(self, model, batch_size):
        self.model, self.eng, self.B = model, model.engine, int(batch_size)
        self.state = self.eng.new_state(self.B)

    def reset_states(self):
        for k in ('h_enc', 'c_enc', 'h_dec', 'c_dec'):
            self.state[k].zero_()


class WEncoder(_Stateful):
    """[w_mean, w_log_var] of a whole window (Flatten -> hW -> Wargs)."""

    def __init__(self, model, seq_length, batch_size):
        super().__init__(model, batch_size)
        if seq_length != self.eng.cfg['T']:
            raise ValueError("hW consumes the flattened window: seq_length must be %d" % self.eng.cfg['T'])
        if batch_size > self.eng.B:
            raise ValueError("sub-model batch %d exceeds the engine's batch %d" % (batch_size, self.eng.B))

    def predict(self, x):
        e, B = self.eng, self.B
        C1 = e.cfg['C'] - 1
        e.encode_w(_dev(x, e.device, (B, e.cfg['T'] * e.cfg['D'])), B)
        wa = e.wargs[:B].cpu().numpy()
        return [wa[:, :C1].copy(), wa[:, C1:].copy()]


class ZEncoder(_Stateful):
    def __init__(self, model, batch_size, emulate_fresh_encoder=False, seed=None):
        super().__init__(model, batch_size)
        self.rec_name = 'encoder_h'
        self._saved = None
        if emulate_fresh_encoder:
            # reference :122-125: a new LSTM layer named encoder_h with default initialisers
            rng = np.random.default_rng(seed)
            H = self.eng.cfg['H']
            shp = dict(vrnn_param_shapes(self.eng.cfg))
            bias = np.zeros(4 * H, np.float32)
            bias[H:2 * H] = 1.0
            self._fresh = dict(kernel=glorot_uniform(rng, shp['encoder_h/kernel']),
                               recurrent_kernel=orthogonal(rng, shp['encoder_h/recurrent_kernel']), bias=bias)
        else:
            self._fresh = None

    def predict(self, xs):
        e, B, st = self.eng, self.B, self.state
        x, w = xs
        L = e.cfg['L']
        if self._fresh is not None:
            lay = self.model.get_layer('encoder_h')
            trained = lay.get_weights()
            lay.set_weights([self._fresh['kernel'], self._fresh['recurrent_kernel'], self._fresh['bias']])
        e.enc_step(_dev(x, e.device, (B, e.cfg['D'])), _dev(w, e.device, (B, e.cfg['C'])), st)
        za = st['zargs'].cpu().numpy()
        if self._fresh is not None:
            lay.set_weights(trained)
        return [za[:, None, :L].copy(), za[:, None, L:].copy()]


class Decoder(_Stateful):
    def predict(self, xs):
        e, B, st = self.eng, self.B, self.state
        if e.cfg['use_x_prev']:
            z, xp, w = xs
            xp = _dev(xp, e.device, (B, e.cfg['D']))
        else:
            (z, w), xp = xs, None
        e.dec_step(_dev(z, e.device, (B, e.cfg['L'])), xp, _dev(w, e.device, (B, e.cfg['C'])), st)
        return st['xhat'].cpu().numpy()[:, None, :].copy()


class Encoder:
    """`encoder` of get_model: X -> [Z_mean, Z_log_var, W] (cl_vrnn/model.py:266)."""

    def __init__(self, model):
        self.model = model

    def predict(self, X, batch_size=None):
        m = self.model
        L = m.engine.cfg['L']
        xs = [X, np.zeros_like(np.asarray(X))] if m.use_x_prev else X
        _, W, _, Zargs = m.predict(xs)
        return [Zargs[..., :L], Zargs[..., L:], W]


def generate_samples_device(model, x_seeds, nsteps, w_vals, seed=0, z_prior=False):
    """Batched, device-resident counterpart of generate_sample: N seeds at once, the whole frame loop as
    replays of one captured hipGraph, Philox noise instead of np.random (so the draws differ from the numpy
    path, the distribution does not).  x_seeds [N,S,88] (S >= 0 teacher-forced frames), w_vals [N,C].
    Returns the free-running part [N,nsteps,88] as a numpy array, like generate_sample does per seed."""
    e = model.engine
    xs = torch.as_tensor(np.ascontiguousarray(np.asarray(x_seeds), dtype=np.float32), device=e.device)
    w = torch.as_tensor(np.ascontiguousarray(np.asarray(w_vals), dtype=np.float32), device=e.device)
    return e.generate(xs, w, int(nsteps), seed=int(seed), z_prior=z_prior).cpu().numpy().astype(np.float64)


def make_w_encoder(model, original_dim, n_classes, seq_length=1, batch_size=1):
    return WEncoder(model, seq_length, batch_size)


def make_z_encoder(model, original_dim, n_classes, latent_dims, seq_length=1, batch_size=1, stateful=True,
                   emulate_fresh_encoder=False):
    if seq_length != 1:
        raise ValueError("the step z-encoder advances one frame per predict() call (seq_length=1)")
    return ZEncoder(model, batch_size, emulate_fresh_encoder)


def make_decoder(model, original_dim, intermediate_dim, latent_dim, n_classes, use_x_prev, seq_length=1, batch_size=1,
                 stateful=True):
    if seq_length != 1:
        raise ValueError("the step decoder advances one frame per predict() call (seq_length=1)")
    if bool(use_x_prev) != bool(model.engine.cfg['use_x_prev']):
        raise ValueError("use_x_prev does not match the model the decoder is taken from")
    return Decoder(model, batch_size)


def get_model(batch_size, original_dim, intermediate_dim, latent_dim, seq_length, n_classes, use_x_prev, optimizer,
              class_weight=1.0, kl_weight=1.0, dropout=0.0, w_kl_weight=1.0, w_log_var_prior=0.0, seed=None,
              device='cuda:0', gate_act='hard_sigmoid'):
    """-> (model, encoder).  dropout (cl_vrnn/model.py:164,198,227: `LSTM(..., dropout=dropout)` for both LSTMs; 0 at every
    call site of the reference, cl_vrnn/train.py:46): input dropout in the training steps of fit(), none in validation,
    predict() and generation (Keras' learning phase); > 0 takes the engine's generic kernel chain (VrnnEngine._forward_dropout)."""
    cfg = dict(D=int(original_dim), H=int(intermediate_dim), L=int(latent_dim), T=int(seq_length), C=int(n_classes),
               use_x_prev=bool(use_x_prev), class_weight=get_value(class_weight), kl_weight=get_value(kl_weight),
               w_kl_weight=get_value(w_kl_weight), w_log_var_prior=float(w_log_var_prior), gate_act=gate_act,
               dropout=float(dropout))
    eng = VrnnEngine(cfg, batch_size, device)
    eng.P.set_weights(init_weights(eng.P.logical, cfg, seed=seed))
    model = ClVrnnModel(eng, optimizer, kl_weight, w_kl_weight, class_weight, bool(use_x_prev), seed=seed)
    return model, Encoder(model)


def load_model(model_file, batch_size=None, seq_length=None, optimizer='adam'):
    margs = json.load(open(model_file.replace('.h5', '.json')))
    optimizer = margs['optimizer'] if optimizer is None else optimizer
    batch_size = margs['batch_size'] if batch_size is None else batch_size
    seq_length = margs['seq_length'] if seq_length is None else seq_length
    model, enc_model = get_model(batch_size, margs['original_dim'], margs['intermediate_dim'], margs['latent_dim'],
                                 seq_length, margs['n_classes'], margs['use_x_prev'], optimizer, margs['class_weight'])
    model.load_weights(model_file)
    return model, enc_model, margs
