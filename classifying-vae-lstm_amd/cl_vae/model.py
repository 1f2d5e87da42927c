"""Classifying VAE (cl_vae) on the MI355X HIP path.

Same module surface as the reference's code/cl_vae/model.py: get_model (:130-224), load_model
(:226-239), make_w_encoder (:76-85), make_z_encoder (:87-102), make_decoder (:104-128),
generate_sample (:9-42), sample_x / sample_w / sample_z (:44-74).  Python 3 cannot unpack tuple
parameters, so `(latent_dim_0, latent_dim)` and `(class_dim_0, class_dim)` are passed as 2-tuples.

The graph the reference builds out of Keras layers is one fixed chain of HIP kernels here
(engine.VaeEngine); the returned objects only expose the Keras methods the scripts call.
The numpy samplers are host code in the reference too; they live in clvae_amd.sampling and consume np.random in the
reference's order.
"""
import json

import numpy as np
import torch

from .. import ops, sampling
from ..engine import VaeEngine
from ..initializers import init_weights
from ..keras_like import Layer, Model, get_value


# --------------------------------------------------------------------------- #
# host-side sampling (clvae_amd.sampling holds the shared pieces; these are cl_vae's bindings)
# --------------------------------------------------------------------------- #
def sample_x(x_mean):
    """x ~ Bernoulli(x_mean) as 0./1. (reference :44-45: one uniform per entry of the squeezed vector)"""
    return sampling.draw_frame(x_mean, flat=True)


def sample_w(args, nsamps=1, nrm_samp=False, add_noise=True):
    """Label sample from (w_mean, w_log_var) (reference :47-66)."""
    w_mean, w_log_var = args
    return sampling.logistic_normal(w_mean, w_log_var, nsamps, normal_only=nrm_samp, add_noise=add_noise,
                                    transpose_eps=True, mute_through_scale=True)


def sample_z(args, nsamps=1):
    """Latent sample from (z_mean, z_log_var) (reference :68-74)."""
    return sampling.gaussian(args[0], args[1], nsamps)


def generate_sample(dec_model, w_enc_model, z_enc_model, x_seed, nsteps, w_val=None, use_z_prior=False,
                    do_reset=True, w_sample=False, use_x_prev=False):
    """`nsteps` frames after the seed frame (reference :9-42): w once (given, or inferred from the seed), then per
    frame z from the last frame and w, x ~ Bernoulli(decoder(w, z[, history])).  The decoder's history input lags the
    encoder's input by one frame: at step t the encoder sees x_{t-1}, the decoder x_{t-2} (x_seed for both at t = 0)."""
    loop = sampling.HostFrameLoop(dec_model, w_enc_model, z_enc_model, sample_x, sample_w, sample_z)
    x_in = hist = x_seed[None, :]
    w = w_val if w_val is not None else loop.label([x_in], add_noise=w_sample)
    frames = np.zeros((nsteps, x_seed.shape[0]))
    for t in range(nsteps):
        z = loop.latent(x_in, w, from_prior=use_z_prior)
        frames[t] = x_t = loop.frame([w, z, hist] if use_x_prev else [w, z])
        hist, x_in = x_in, x_t
    return frames


def generate_samples_device(model, x_seeds, nsteps, w_vals, seed=0, use_z_prior=False):
    """N sequences at once with the frame loop on the device (VaeEngine.generate: one captured hipGraph replayed per
    frame, Philox noise instead of np.random: same distribution, different draws).  x_seeds [N,D], w_vals [N,C];
    returns [N,nsteps,D] float64 like generate_sample does per sequence."""
    e = model.engine
    t = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32), device=e.device)
    return e.generate(t(x_seeds), t(w_vals), int(nsteps), seed=int(seed), z_prior=use_z_prior).cpu().numpy() \
        .astype(np.float64)


# --------------------------------------------------------------------------- #
# models
# --------------------------------------------------------------------------- #
def _dev(a, dev, shape=None):
    t = torch.as_tensor(np.ascontiguousarray(np.asarray(a), dtype=np.float32), device=dev)
    return t if shape is None else t.reshape(shape)


class ClVaeModel(Model):
    output_names = ('x_decoded_mean', 'w', 'w2', 'z_args')
    acc_name = 'w_acc'

    def __init__(self, engine, optimizer, kl_weight, w_kl_weight, class_weight, use_x_prev, seed=None):
        super().__init__(engine, optimizer, kl_weight, w_kl_weight, class_weight, seed)
        self.use_x_prev = use_x_prev
        self.inputs = ['x', 'history'] if use_x_prev else ['x']
        # model.layers in the topological order Keras reports for cl_vae/model.py:136-209
        L = lambda n, w=(), c='Dense': Layer(n, self, w, c)
        kb = ('kernel', 'bias')
        hidden = engine.cfg['H'] > 0       # intermediate_dim == 0: no `h` / `decoder_h` layers (reference :165-167,188)
        order = [L('x', c='InputLayer'), L('h_w', kb), L('w_mean', kb), L('w_log_var', kb), L('w', c='Lambda'),
                 L('concatenate_1', c='Concatenate')] + ([L('h', kb)] if hidden else []) + [L('z_mean', kb), L('z_log_var', kb)]
        if use_x_prev:
            order += [L('history', c='InputLayer')]
        order += [L('z', c='Lambda')]
        if use_x_prev:
            order += [L('concatenate_2', c='Concatenate')]
        order += [L('concatenate_3' if use_x_prev else 'concatenate_2', c='Concatenate')] \
            + ([L('decoder_h', kb)] if hidden else []) \
            + [L('x_decoded_mean', kb), L('w2', c='Lambda'), L('z_args', c='Concatenate')]
        self.layers = order

    def _split_inputs(self, x, y):
        if self.use_x_prev:
            cur, hist = x[0], x[1]
        else:
            cur, hist = (x[0] if isinstance(x, (list, tuple)) else x), None
        return np.asarray(cur), (None if hist is None else np.asarray(hist)), np.asarray(y[1])

    def predict(self, x, batch_size=None, verbose=0):
        """[x_decoded_mean, w, w2, z_args] with freshly drawn noise, in chunks of the model's batch size."""
        eng = self.engine
        cur, hist, _ = self._split_inputs(x, [None, np.zeros((len(np.asarray(x[0] if self.use_x_prev else x)), eng.cfg['C']))])
        B, dev = eng.B, eng.device
        n = cur.shape[0]
        if n % B:
            raise ValueError("predict needs a multiple of the fixed batch size %d" % B)
        ts = self._train_step()
        outs = [[], [], [], []]
        for b0 in range(0, n, B):
            ts.X.copy_(_dev(cur[b0:b0 + B], dev))
            if hist is not None:
                ts.Xp.copy_(_dev(hist[b0:b0 + B], dev))
            ts.draw_noise(stream_offset=1000 + b0 // B)
            eng.forward(ts.X, ts.Xp, ts.eps_w, ts.eps_z)
            outs[0].append(eng.x_hat().cpu().numpy())
            w = eng.w.cpu().numpy()
            outs[1].append(w); outs[2].append(w + 1e-10)
            outs[3].append(eng.zargs.cpu().numpy())
        return [np.concatenate(o) for o in outs]


class _SubModel:
    """Batch-sized inference graph sharing the trained layers (the reference re-uses layer objects)."""

    def __init__(self, model, batch_size):
        self.model = model
        self.eng = model.engine
        self.B = batch_size
        if batch_size > self.eng.B:
            raise ValueError("sub-model batch %d exceeds the engine's batch %d" % (batch_size, self.eng.B))

    def reset_states(self):
        pass


class WEncoder(_SubModel):
    def predict(self, x):
        e, B = self.eng, self.B
        C1 = e.cfg['C'] - 1
        e.encode_w(_dev(x, e.device, (B, e.cfg['D'])), B)
        wa = e.wargs[:B].cpu().numpy()
        return [wa[:, :C1].copy(), wa[:, C1:].copy()]


class ZEncoder(_SubModel):
    def predict(self, xs):
        e, B = self.eng, self.B
        x, w = xs
        L = e.cfg['L']
        e.encode_z(_dev(x, e.device, (B, e.cfg['D'])), _dev(w, e.device, (B, e.cfg['C'])), B)
        za = e.zargs[:B].cpu().numpy()
        return [za[:, :L].copy(), za[:, L:].copy()]


class Decoder(_SubModel):
    def predict(self, xs):
        e, B = self.eng, self.B
        w, z = xs[0], xs[1]
        xp = _dev(xs[2], e.device, (B, e.cfg['D'])) if e.cfg['use_x_prev'] else None
        e.decode(_dev(w, e.device, (B, e.cfg['C'])), _dev(z, e.device, (B, e.cfg['L'])), xp, B, act=ops.ACT_SIGMOID)
        return e.logits[:B].cpu().numpy().copy()


class EncModel:
    """enc_model of get_model: x -> [z_mean, w_mean] (cl_vae/model.py:211-212,220-223)."""

    def __init__(self, model):
        self.model = model

    def predict(self, x, batch_size=None):
        m = self.model
        e = m.engine
        L, C1 = e.cfg['L'], e.cfg['C'] - 1
        z_means, w_means = [], []
        cur = np.asarray(x[0] if m.use_x_prev else (x[0] if isinstance(x, (list, tuple)) else x))
        B, dev = e.B, e.device
        if cur.shape[0] % B:
            raise ValueError("predict needs a multiple of the fixed batch size %d" % B)
        ts = m._train_step()
        for b0 in range(0, cur.shape[0], B):
            ts.X.copy_(_dev(cur[b0:b0 + B], dev))
            ts.draw_noise(stream_offset=2000 + b0 // B)
            e.forward(ts.X, ts.Xp, ts.eps_w, ts.eps_z)
            z_means.append(e.zargs[:, :L].cpu().numpy())
            w_means.append(e.wargs[:, :C1].cpu().numpy())
        return [np.concatenate(z_means), np.concatenate(w_means)]


def make_w_encoder(model, original_dim, batch_size=1):
    return WEncoder(model, batch_size)


def make_z_encoder(model, original_dim, class_dim, latent_dims, batch_size=1):
    return ZEncoder(model, batch_size)


def make_decoder(model, latent_dims, class_dim, original_dim=88, use_x_prev=False, batch_size=1):
    if bool(use_x_prev) != bool(model.engine.cfg['use_x_prev']):
        raise ValueError("use_x_prev does not match the model the decoder is taken from")
    return Decoder(model, batch_size)


def get_model(batch_size, original_dim, latent_dims, class_dims, optimizer, class_weight=1.0, kl_weight=1.0,
              use_x_prev=False, w_kl_weight=1.0, w_log_var_prior=0.0, seed=None, device='cuda:0', bf16=False):
    """-> (model, enc_model).  latent_dims = (latent_dim_0, latent_dim); class_dims = (class_dim_0, class_dim).
    bf16: the Dense products of the fused training step run on the bf16 matrix cores (engine cfg 'bf16')."""
    latent_dim_0, latent_dim = latent_dims
    class_dim_0, class_dim = class_dims
    cfg = dict(D=int(original_dim), H=int(latent_dim_0), L=int(latent_dim), Hc=int(class_dim_0), C=int(class_dim),
               use_x_prev=bool(use_x_prev), class_weight=get_value(class_weight), kl_weight=get_value(kl_weight),
               w_kl_weight=get_value(w_kl_weight), w_log_var_prior=float(w_log_var_prior), bf16=bool(bf16))
    eng = VaeEngine(cfg, batch_size, device)
    eng.P.set_weights(init_weights(eng.P.logical, cfg, seed=seed))
    model = ClVaeModel(eng, optimizer, kl_weight, w_kl_weight, class_weight, bool(use_x_prev), seed=seed)
    return model, EncModel(model)


def load_model(model_file, optimizer='adam', batch_size=1, no_x_prev=False):
    """Rebuild from <run>.json and load <run>.h5 (the reference's work-around for Lambda layers in YAML)."""
    margs = json.load(open(model_file.replace('.h5', '.json')))
    batch_size = margs['batch_size'] if batch_size is None else batch_size
    if no_x_prev or 'use_x_prev' not in margs:
        margs['use_x_prev'] = False
    model, enc_model = get_model(batch_size, margs['original_dim'], (margs['intermediate_dim'], margs['latent_dim']),
                                 (margs['intermediate_class_dim'], margs['n_classes']), optimizer,
                                 margs['class_weight'], use_x_prev=margs['use_x_prev'])
    model.load_weights(model_file)
    return model, enc_model, margs
