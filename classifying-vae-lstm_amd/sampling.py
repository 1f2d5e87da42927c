"""Host-side sampling: the numpy draws and the frame-by-frame generation protocol of both models.

What the reference defines twice (cl_vae/model.py:9-74 and cl_vrnn/model.py:9-96) is one set of building blocks
here; `cl_vae.model` and `cl_vrnn.model` bind them with the small differences between the two files:

  * `draw_frame`      x_t ~ Bernoulli(x_mean):   `u <= x_mean` as float64, u from `np.random.rand`
  * `logistic_normal` the label sample:          softmax([mean + exp(log_var/2) * eps, 0])
  * `gaussian`        the latent sample:         mean + exp(log_var/2) * eps
  * `one_hot_draw`    a categorical draw from w  (cl_vrnn only)
  * `HostFrameLoop`   generate_sample's loop over three predict()-style sub-models

Everything draws from the GLOBAL `np.random` state, in the reference's order and with the reference's array shapes, so a
seeded run consumes the generator identically (pinned by tests/golden/g2_g3_samplers.npz, which was produced by the
reference's own functions).  The device-side generation (engine.generate, Philox noise) does not go through this module.
"""
import numpy as np


def draw_frame(x_mean, flat):
    """1.0 * (u <= x_mean).  flat=True draws len(squeeze(x_mean)) uniforms (cl_vae/model.py:45), otherwise one per
    element of squeeze(x_mean) (cl_vrnn/model.py:63); the comparison broadcasts against the unsqueezed x_mean."""
    sq = np.squeeze(x_mean)
    u = np.random.rand(len(sq)) if flat else np.random.rand(*sq.shape)
    return (u <= x_mean) * 1.0


def _eps_like(mean, nsamps, flatten_single):
    """Standard normal noise for `nsamps` samples of a tensor shaped like `mean`.  A single sample of a label vector is
    drawn as ONE row over all entries (both files: `randn(1, mean.size)`), of a latent as squeeze(mean)."""
    if nsamps != 1:
        return np.random.randn(nsamps, *mean.shape)
    return np.random.randn(1, mean.size) if flatten_single else np.random.randn(*mean.shape)


def _softmax_with_zero_logit(a):
    """softmax over the last axis of [a, 0]: exp / sum(exp), no max shift (same floating-point operations as the
    reference's hstack/dstack + exp/sum lines)."""
    e = np.exp(np.concatenate([a, np.zeros(a.shape[:-1] + (1,))], axis=-1))
    return e / e.sum(axis=-1, keepdims=True)


def logistic_normal(mean, log_var, nsamps=1, normal_only=False, add_noise=True, transpose_eps=False,
                    mute_through_scale=False):
    """The label sample w.  transpose_eps: cl_vae flips a noise matrix whose transpose has the mean's shape (:53-54).
    add_noise=False multiplies the noise by zero AFTER it is drawn (the draw still advances the generator);
    mute_through_scale keeps cl_vae's form `0 * exp(log_var/2) * eps` (a NaN where the scale overflows), cl_vrnn's is
    `0 * eps`."""
    eps = _eps_like(mean, nsamps, flatten_single=True)
    if transpose_eps and eps.T.shape == mean.shape:
        eps = eps.T
    if add_noise:
        pre = mean + np.exp(log_var / 2) * eps
    elif mute_through_scale:
        pre = mean + 0 * np.exp(log_var / 2) * eps
    else:
        pre = mean + 0 * eps
    return pre if normal_only else _softmax_with_zero_logit(pre)


def gaussian(mean, log_var, nsamps=1):
    """z = mean + exp(log_var/2) * eps with eps shaped like squeeze(mean) (times nsamps)."""
    eps = _eps_like(np.squeeze(mean), nsamps, flatten_single=False)
    return mean + np.exp(log_var / 2) * eps


def one_hot_draw(w):
    """One categorical draw from the (renormalised) label vector, as a one-hot vector."""
    out = np.zeros(w.shape)
    out[np.random.choice(len(w), p=w / w.sum())] = 1.
    return out


class HostFrameLoop:
    """generate_sample as a protocol over three sub-models with Keras' predict():
         w_enc(frames) -> (w_mean, w_log_var);  z_enc([x, w]) -> (z_mean, z_log_var);  dec([...]) -> x_mean.
    `label()` settles w once, `frame()` produces one frame.  The sub-models may be the device-backed ones of this
    package or anything else with predict() (the golden tests drive it with recording stubs)."""

    def __init__(self, dec_model, w_enc_model, z_enc_model, sample_x, sample_w, sample_z):
        self.dec, self.w_enc, self.z_enc = dec_model, w_enc_model, z_enc_model
        self.sample_x, self.sample_w, self.sample_z = sample_x, sample_w, sample_z

    def reset(self):
        for m in (self.dec, self.w_enc, self.z_enc):      # this order: cl_vrnn/model.py:22-24
            m.reset_states()

    def label(self, windows, add_noise):
        """Mean over the label samples of each window (one window: that sample itself)."""
        ws = [self.sample_w(self.w_enc.predict(win), add_noise=add_noise) for win in windows]
        return ws[0] if len(ws) == 1 else np.vstack(ws).mean(axis=0)[None, :]

    def latent(self, x, w, from_prior=False):
        stats = self.z_enc.predict([x, w])
        if from_prior:
            stats = tuple(0 * s for s in stats)
        return self.sample_z(tuple(stats))

    def frame(self, decoder_inputs):
        return self.sample_x(self.dec.predict(decoder_inputs))
