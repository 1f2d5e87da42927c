#!/usr/bin/env python
"""G4-independent: a SECOND derivation of the cl_vae / cl_vrnn training losses and gradients, stored as a fixture.

Why: every arithmetic test compares the HIP kernels with oracle/clvae_oracle.py (numpy, hand-derived backward pass), and the
G4 fixture of make_golden.py holds that oracle's own numbers.  Keras / TensorFlow cannot run in this image, so the oracle
cannot be pinned to the reference itself; what CAN be done is to derive the same quantities a different way, by someone
reading only the reference's model definitions.  This script

  * imports nothing from oracle/ (and nothing from the package): it is written from /root/reference/code/cl_vae/model.py
    and /root/reference/code/cl_vrnn/model.py (line numbers below) plus the Keras 2.0 semantics those lines call;
  * builds the graphs in float64 torch and takes every gradient with torch.autograd (no hand-written backward);
  * writes inputs (weights, frames, labels, noise) and outputs (loss terms, per-note logits, every gradient tensor) to
    tests/golden/g4_independent.npz.

tests/test_oracle_golden.py holds the oracle to these numbers on CPU, tests/test_gpu_models.py the HIP kernels on the GPU.
Runs in the build container only (real JSB frames come from /root/reference/data/input through the reference's own
utils/pianoroll.py, imported like make_golden.py imports it).  Only arrays are stored.

Keras 2.0.x (TensorFlow backend) semantics used, as the reference's requirements pin them:
  Dense(units, activation)            y = act(x . kernel + bias)                        [keras/layers/core.py]
  LSTM (implementation 0/1)           gates from x . kernel[:, i|f|c|o] + h . recurrent_kernel[...] + bias, blocks in the
                                      order i, f, c, o; recurrent_activation = hard_sigmoid = clip(0.2 x + 0.5, 0, 1);
                                      activation = tanh; c' = f c + i tanh(z_c); h' = o tanh(c')   [keras/layers/recurrent.py]
  losses.binary_crossentropy          mean over the last axis of K.binary_crossentropy(output, target): the TF backend
                                      clips output to [eps, 1 - eps] (eps = 1e-7, as float32), takes log(p / (1 - p)) and
                                      applies sigmoid_cross_entropy_with_logits = max(l, 0) - l y + log(1 + exp(-|l|))
  losses.categorical_crossentropy     output / sum(output, -1), clipped to [eps, 1 - eps], -sum(target log(output), -1)
  Model.compile(loss dict, weights)   total = sum_k weight_k * mean over every sample (and timestep) of loss_k
  metrics 'accuracy' on a one-hot     mean(argmax(y_true) == argmax(y_pred))
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, import_reference_pianoroll  # noqa: E402  (the pianoroll loader only: no oracle import)

torch.set_default_dtype(torch.float64)
F32 = np.float32
EPS32 = float(F32(1e-7))                      # K.epsilon() as the float32 graph holds it
P_HI32 = float(F32(1.0) - F32(1e-7))          # 1 - eps in float32 = 1 - 2^-23


def hard_sigmoid(x):
    return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)


def keras_binary_crossentropy_sum(logits, target):
    """original_dim * binary_crossentropy(x, sigmoid(logits)) = the SUM over the notes (cl_vae/model.py:190-191,
    cl_vrnn/model.py:241-242).  The float32 graph clips p at [float32(1e-7), float32(1 - 1e-7)]; log(p / (1 - p)) of the
    clipped p, evaluated on the logits: p inside the clip gives back the logit itself, p outside gives the constant of
    the clip point (no gradient)."""
    lo = np.log(np.float64(F32(EPS32) / (F32(1.0) - F32(EPS32))))        # float32 quotient, like the graph forms it
    hi = np.log(np.float64(F32(P_HI32) / (F32(1.0) - F32(P_HI32))))      # = log(2^23 - 1)
    l = torch.clamp(logits, float(lo), float(hi))
    return (torch.clamp(l, min=0.0) - l * target + torch.log1p(torch.exp(-l.abs()))).sum(-1)


def keras_categorical_crossentropy(target, output):
    o = output / output.sum(-1, keepdim=True)
    o = torch.clamp(o, 1e-7, 1.0 - 1e-7)
    return -(target * torch.log(o)).sum(-1)


def logit_normal(mean, log_var, eps):
    """w_sampling (cl_vae/model.py:147-157) / sampling_w (cl_vrnn/model.py:183-191): softmax of [mean + exp(lv/2) eps, 0]"""
    s = mean + torch.exp(log_var / 2) * eps
    s0 = torch.cat([s, torch.zeros_like(s[..., :1])], -1)
    e = torch.exp(s0)
    return e / e.sum(-1, keepdim=True)


def w_kl(mean, log_var, prior):
    """w_kl_loss (cl_vae/model.py:204-208, cl_vrnn/model.py:247-252)"""
    vs = 1 - prior + log_var - torch.exp(log_var) / np.exp(prior) - mean ** 2 / np.exp(prior)
    return -0.5 * vs.sum(-1)


def z_kl(mean, log_var):
    """kl_loss (cl_vae/model.py:193-196, cl_vrnn/model.py:236-239)"""
    return -0.5 * (1 + log_var - mean ** 2 - torch.exp(log_var)).sum(-1)


def lstm(x_seq, kernel, rec, bias):
    """keras.layers.LSTM(return_sequences=True), zero initial state: x_seq [B,T,I] -> h [B,T,H]"""
    B, T, _ = x_seq.shape
    H = rec.shape[0]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    out = []
    for t in range(T):
        z = x_seq[:, t] @ kernel + h @ rec + bias
        i, f = hard_sigmoid(z[:, :H]), hard_sigmoid(z[:, H:2 * H])
        g, o = torch.tanh(z[:, 2 * H:3 * H]), hard_sigmoid(z[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out.append(h)
    return torch.stack(out, 1)


def cl_vae_losses(p, x, xp, w_true, eps_w, eps_z, wts, prior, use_x_prev=True):
    """cl_vae/model.py:130-224 (get_model with latent_dim_0 > 0).  wts = (class_weight, kl_weight, w_kl_weight)."""
    relu = torch.relu
    h_w = relu(x @ p['h_w/kernel'] + p['h_w/bias'])                                   # :139
    w_mean = h_w @ p['w_mean/kernel'] + p['w_mean/bias']                              # :140
    w_log_var = h_w @ p['w_log_var/kernel'] + p['w_log_var/bias']                     # :141
    w = logit_normal(w_mean, w_log_var, eps_w)                                        # :144-157
    xw = torch.cat([x, w], -1)                                                        # :160
    h = relu(xw @ p['h/kernel'] + p['h/bias'])                                        # :162
    z_mean = h @ p['z_mean/kernel'] + p['z_mean/bias']                                # :163
    z_log_var = h @ p['z_log_var/kernel'] + p['z_log_var/bias']                       # :164
    z = z_mean + torch.exp(z_log_var / 2) * eps_z                                     # :170-174
    xpz = torch.cat([xp, z], -1) if use_x_prev else z                                 # :177-180
    wz = torch.cat([w, xpz], -1)                                                      # :181
    h_dec = relu(wz @ p['decoder_h/kernel'] + p['decoder_h/bias'])                    # :184-185
    logits = h_dec @ p['x_decoded_mean/kernel'] + p['x_decoded_mean/bias']            # :182,186 (sigmoid inside the loss)
    C = w.shape[-1]
    terms = dict(vae=keras_binary_crossentropy_sum(logits, x).mean(),                 # :190-191
                 kl_z=z_kl(z_mean, z_log_var).mean(),                                 # :193-196
                 w_rec=((C - 1) * keras_categorical_crossentropy(w_true, w + 1e-10)).mean(),     # :198-199, :210
                 kl_w=w_kl(w_mean, w_log_var, prior).mean())                          # :204-208
    cw, kw, wkw = wts
    terms['total'] = terms['vae'] + wkw * terms['kl_w'] + cw * terms['w_rec'] + kw * terms['kl_z']      # :218-221
    terms['acc'] = (w.argmax(-1) == w_true.argmax(-1)).double().mean()
    return terms, logits


def cl_vrnn_losses(p, X, Xp, w_true, eps_W, eps_Z, wts, prior, use_x_prev=True):
    """cl_vrnn/model.py:164-267 (get_model).  X, Xp [B,T,D]."""
    B, T, D = X.shape
    hW = torch.relu(X.reshape(B, T * D) @ p['hW/kernel'] + p['hW/bias'])              # :174
    Wargs = hW @ p['Wargs/kernel'] + p['Wargs/bias']                                  # :175
    C1 = Wargs.shape[-1] // 2
    W_mean, W_log_var = Wargs[:, :C1], Wargs[:, C1:]                                  # :176-181
    W = logit_normal(W_mean, W_log_var, eps_W)                                        # :183-192
    Wrep = W[:, None, :].expand(B, T, W.shape[-1])                                    # RepeatVector(seq_length)
    XW = torch.cat([X, Wrep], -1)                                                     # :194
    enc = lstm(XW, p['encoder_h/kernel'], p['encoder_h/recurrent_kernel'], p['encoder_h/bias'])       # :197-200
    Z_mean = enc @ p['Z_mean/kernel'] + p['Z_mean/bias']                              # :201-210
    Z_log_var = enc @ p['Z_log_var/kernel'] + p['Z_log_var/bias']
    Z = Z_mean + torch.exp(Z_log_var / 2) * eps_Z                                     # :213-217
    XpZ = torch.cat([Xp, Z], -1) if use_x_prev else Z                                 # :219-222
    XpZ = torch.cat([XpZ, Wrep], -1)                                                  # :223
    dec = lstm(XpZ, p['decoder_h/kernel'], p['decoder_h/recurrent_kernel'], p['decoder_h/bias'])      # :226-229
    logits = dec @ p['X_decoded_mean/kernel'] + p['X_decoded_mean/bias']              # :230-235
    C = W.shape[-1]
    terms = dict(vae=keras_binary_crossentropy_sum(logits, X).mean(),                 # :241-242 (mean over B and T)
                 kl_z=z_kl(Z_mean, Z_log_var).mean(),                                 # :236-239
                 w_rec=((C - 1) * keras_categorical_crossentropy(w_true, W + 1e-10)).mean(),       # :244-245, :255
                 kl_w=w_kl(W_mean, W_log_var, prior).mean())                          # :247-252
    cw, kw, wkw = wts
    terms['total'] = terms['vae'] + wkw * terms['kl_w'] + cw * terms['w_rec'] + kw * terms['kl_z']      # :262-265
    terms['acc'] = (W.argmax(-1) == w_true.argmax(-1)).double().mean()
    return terms, logits, enc, dec


def glorot(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, shape).astype(F32)


def run(losses_fn, params, inputs, extra):
    p = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in params.items()}
    out = losses_fn(p, *[torch.tensor(np.asarray(a, np.float64)) for a in inputs], **extra)
    terms, logits = out[0], out[1]
    terms['total'].backward()
    res = {'loss/' + k: np.array(float(v)) for k, v in terms.items()}
    res['logits'] = logits.detach().numpy()
    for k, v in p.items():
        res['g/' + k] = v.grad.numpy().astype(F32 if v.numel() > 20000 else np.float64)
    if len(out) > 2:
        res['enc_h'] = out[2].detach().numpy().astype(F32)
        res['dec_h'] = out[3].detach().numpy().astype(F32)
    return res


def main():
    ref = import_reference_pianoroll()
    rng = np.random.default_rng(20261003)
    out = {}

    # ---- cl_vae: 24 frames of JSB Chorales_Cs, --use_x_prev, latent 4, 2 classes (BASELINE config 1's model) ----------
    P = ref.PianoData(os.path.join(REF, 'data/input/JSB Chorales_Cs.pickle'), batch_size=100, seq_length=1,
                      step_length=1, return_y_next=True, squeeze_x=True, squeeze_y=True)
    B, D, H, L, C = 24, 88, 88, 4, 2
    sel = rng.permutation(len(P.y_train))[:B]
    x, xp = P.y_train[sel], P.x_train[sel]                     # cl_vae/train.py:58-60: x = the frame, history = the one before
    wt = np.eye(C)[np.asarray(P.train_song_keys)[sel]]
    ew, ez = rng.standard_normal((B, C - 1)).astype(F32), rng.standard_normal((B, L)).astype(F32)
    shapes = [('h_w', D, H), ('w_mean', H, C - 1), ('w_log_var', H, C - 1), ('h', D + C, H), ('z_mean', H, L),
              ('z_log_var', H, L), ('decoder_h', C + D + L, H), ('x_decoded_mean', H, D)]
    pv = {}
    for name, i, o in shapes:
        pv[name + '/kernel'] = glorot(rng, (i, o))
        pv[name + '/bias'] = (0.1 * rng.standard_normal(o)).astype(F32)
    pv['x_decoded_mean/bias'] += F32(-2.0)                     # piano-roll density; some logits reach the clip region
    pv['x_decoded_mean/kernel'][:, 5] *= F32(40.0)             # one note column far outside the epsilon clip
    wts, prior = (0.8, 0.6, 0.9), 0.2
    res = run(cl_vae_losses, pv, (x, xp, wt, ew, ez), dict(wts=wts, prior=prior))
    out.update({'vae/x': x.astype(np.uint8), 'vae/xp': xp.astype(np.uint8), 'vae/wt': wt, 'vae/ew': ew, 'vae/ez': ez,
                'vae/wts': np.array(wts), 'vae/prior': np.array(prior)})
    out.update({'vae/p/' + k: v for k, v in pv.items()})
    out.update({'vae/' + k: v for k, v in res.items()})

    # ---- cl_vrnn: 6 windows of 8 frames of JSB Chorales_all, 10 classes, latent 2 (BASELINE config 3's model) ------------
    T = 8
    P2 = ref.PianoData(os.path.join(REF, 'data/input/JSB Chorales_all.pickle'), batch_size=200, seq_length=T,
                       step_length=1, return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)
    B, L, C = 6, 2, 10
    sel = rng.permutation(len(P2.y_train))[:B]
    X, Xp = P2.y_train[sel], P2.x_train[sel]                   # cl_vrnn/train.py:51-53
    wt = np.eye(C)[np.asarray(P2.train_song_keys)[sel]]
    eW, eZ = rng.standard_normal((B, C - 1)).astype(F32), rng.standard_normal((B, T, L)).astype(F32)
    pv = {'hW/kernel': glorot(rng, (T * D, D)), 'hW/bias': (0.1 * rng.standard_normal(D)).astype(F32),
          'Wargs/kernel': glorot(rng, (D, 2 * (C - 1))), 'Wargs/bias': (0.1 * rng.standard_normal(2 * (C - 1))).astype(F32)}
    for name, i in (('encoder_h', D + C), ('decoder_h', D + L + C)):
        pv[name + '/kernel'] = glorot(rng, (i, 4 * H))
        pv[name + '/recurrent_kernel'] = (rng.standard_normal((H, 4 * H)) / np.sqrt(H)).astype(F32)
        pv[name + '/bias'] = (0.1 * rng.standard_normal(4 * H)).astype(F32)
    for name, o in (('Z_mean', L), ('Z_log_var', L)):
        pv[name + '/kernel'] = (0.1 * rng.standard_normal((H, o))).astype(F32)          # RandomNormal(stddev=0.1), :202-209
        pv[name + '/bias'] = (0.05 * rng.standard_normal(o)).astype(F32)
    pv['X_decoded_mean/kernel'] = (0.1 * rng.standard_normal((H, D))).astype(F32)
    pv['X_decoded_mean/bias'] = (0.05 * rng.standard_normal(D) - 2.0).astype(F32)
    # note columns whose logits sit beyond, between and just inside the Bernoulli clip points (+15.942 / -16.118)
    pv['X_decoded_mean/bias'][7:13] = np.array([17.0, -17.5, 15.96, -16.0, 16.05, 15.9], F32)
    # larger recurrent gate pre-activations: the hard sigmoid's flat regions are exercised (its gradient is 0 there)
    pv['encoder_h/bias'][:H] += F32(2.0)
    pv['decoder_h/bias'][3 * H:] -= F32(2.0)
    res = run(cl_vrnn_losses, pv, (X, Xp, wt, eW, eZ), dict(wts=wts, prior=prior))
    out.update({'vrnn/X': X.astype(np.uint8), 'vrnn/Xp': Xp.astype(np.uint8), 'vrnn/wt': wt, 'vrnn/eW': eW, 'vrnn/eZ': eZ,
                'vrnn/wts': np.array(wts), 'vrnn/prior': np.array(prior)})
    out.update({'vrnn/p/' + k: v for k, v in pv.items()})
    out.update({'vrnn/' + k: v for k, v in res.items()})
    path = os.path.join(HERE, 'g4_independent.npz')
    np.savez_compressed(path, **out)
    print('G4-independent:', len(out), 'arrays,', os.path.getsize(path) // 1024, 'KiB')
    for m in ('vae', 'vrnn'):
        print(m, {k.split('/')[-1]: float(v) for k, v in out.items() if k.startswith(m + '/loss/')})


if __name__ == '__main__':
    main()
