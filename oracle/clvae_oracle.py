"""numpy restatement of the reference's cl_vae / cl_vrnn model math (TEST ORACLE).

Test infrastructure only -- see oracle/__init__.py for who may import this and
for the parity status ("unpinned at the Keras boundary").

Every function cites the reference file:line it restates (paths relative to
/root/reference/code).  Keras/TF semantics that are not in the reference checkout
are the ones recalled in SURVEY.md Appendix A (marked [K] there).

Conventions
  * arrays are numpy, dtype selectable (float64 default = the "exact" oracle,
    float32 = what Keras' floatx would compute);
  * weights use the Keras layouts: Dense kernel [in, out], bias [out];
    LSTM kernel [in, 4H], recurrent_kernel [H, 4H], bias [4H], gate blocks in
    the order i, f, c, o along the 4H axis;
  * parameters live in an ordered dict  "<layer>/<weight>" -> array.
"""
from __future__ import annotations

import numpy as np

EPS_K = 1e-7          # keras.backend._EPSILON [K]
W2_SHIFT = 1e-10      # cl_vae/model.py:208, cl_vrnn/model.py:255


# --------------------------------------------------------------------------- #
# small helpers
# --------------------------------------------------------------------------- #
def relu(a):
    return np.maximum(a, 0)


def sigmoid(a):
    out = np.empty_like(a)
    pos = a >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-a[pos]))
    e = np.exp(a[~pos])
    out[~pos] = e / (1.0 + e)
    return out


def hard_sigmoid(z):
    """Keras 2.0.0 TF backend: clip(0.2*x + 0.5, 0, 1) [K] (Appendix A.2)."""
    return np.clip(0.2 * z + 0.5, 0.0, 1.0)


def hard_sigmoid_grad(z):
    """d/dz of clip(0.2 z + 0.5, 0, 1); TF's clip passes the gradient at ties."""
    y = 0.2 * z + 0.5
    return np.where((y >= 0.0) & (y <= 1.0), 0.2, 0.0).astype(z.dtype)


def glorot_uniform(rng, shape, dtype):
    fan_in, fan_out = shape[0], shape[1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def orthogonal(rng, shape, dtype):
    a = rng.standard_normal(shape)
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == tuple(shape) else vt
    return q.astype(dtype)


# --------------------------------------------------------------------------- #
# losses (Keras semantics, Appendix A.3)
# --------------------------------------------------------------------------- #
# The comparand is the FLOAT32 Keras path (cl_vae/model.py:190-191, cl_vrnn/model.py:241-242): there
# p = clip(sigmoid(a), float32(1e-7), float32(1 - 1e-7)) and float32(1 - 1e-7) = 1 - 2^-23, so
#   upper logit clip = log((1 - 2^-23) / 2^-23) = log(2^23 - 1) = 15.942385
#   lower logit clip = log(float32(1e-7) / float32(1 - 1e-7)) = -16.118095  (as float32 evaluates the quotient)
# LOGIT_CLIP_EXACT is the exact-arithmetic (float64 Keras) form, +-log((1 - 1e-7) / 1e-7).
LOGIT_CLIP_EXACT = float(np.log((1.0 - EPS_K) / EPS_K))   # 16.118095...
_p_hi = np.float32(1.0) - np.float32(EPS_K)
_p_lo = np.float32(EPS_K)
LOGIT_CLIP_HI = float(np.log(np.float64(_p_hi / (np.float32(1.0) - _p_hi))))      # 15.942385...
LOGIT_CLIP_LO = float(np.log(np.float64(_p_lo / (np.float32(1.0) - _p_lo))))      # -16.118095...


def bce_from_logits_keras(a, y, clip='float32'):
    """sum_j binary_crossentropy with the Keras clip, expressed on the logits.

    Keras: p = clip(sigmoid(a), eps, 1-eps); l = log(p/(1-p));
           bce = max(l,0) - l*y + log(1+exp(-|l|))        [K] A.3
    log(p/(1-p)) of the clipped p equals clip(a, -Lc, +Lc) with
    Lc = log((1-eps)/eps), so the loss is softplus(l) - l*y on l = clip(a).
    Restates vae_loss = original_dim * mean_j(...) = sum_j(...)
    (cl_vae/model.py:190-191, cl_vrnn/model.py:241-242).
    clip='float32' (default): the clip points of the float32 Keras path (asymmetric, see above);
    clip='exact': +-log((1-eps)/eps).
    Returns (per-row loss summed over the last axis, dloss/da elementwise).
    """
    lo, hi = (LOGIT_CLIP_LO, LOGIT_CLIP_HI) if clip == 'float32' else (-LOGIT_CLIP_EXACT, LOGIT_CLIP_EXACT)
    l = np.clip(a, lo, hi)
    sp = np.maximum(l, 0) + np.log1p(np.exp(-np.abs(l)))
    loss = (sp - l * y).sum(axis=-1)
    inside = (a >= lo) & (a <= hi)
    grad = np.where(inside, sigmoid(l) - y, 0.0).astype(a.dtype)
    return loss, grad


def cce_keras(w, onehot, scale):
    """scale * categorical_crossentropy(onehot, w + 1e-10) with Keras' renormalise + clip.

    cl_vae/model.py:198-199,208; cl_vrnn/model.py:244-245,255; [K] A.3.
    Returns (loss [B], dloss/dw [B,C]).
    """
    q = w + W2_SHIFT
    S = q.sum(axis=-1, keepdims=True)
    n = q / S
    nc = np.clip(n, EPS_K, 1.0 - EPS_K)
    loss = -scale * (onehot * np.log(nc)).sum(axis=-1)
    inside = (n >= EPS_K) & (n <= 1.0 - EPS_K)
    dn = np.where(inside, -scale * onehot / nc, 0.0)
    dq = (dn - (dn * n).sum(axis=-1, keepdims=True)) / S
    return loss, dq.astype(w.dtype)


def kl_gauss(mean, log_var):
    """-0.5*sum(1 + lv - mean^2 - exp(lv))  (cl_vae/model.py:193-196; cl_vrnn/model.py:236-239)."""
    loss = -0.5 * (1 + log_var - mean ** 2 - np.exp(log_var)).sum(axis=-1)
    return loss, mean.copy(), (-0.5 * (1 - np.exp(log_var))).astype(mean.dtype)


def kl_w_prior(mean, log_var, prior):
    """w_kl_loss (cl_vae/model.py:202-206; cl_vrnn/model.py:247-252)."""
    ep = np.exp(prior)
    vs = 1 - prior + log_var - np.exp(log_var) / ep - mean ** 2 / ep
    loss = -0.5 * vs.sum(axis=-1)
    dmean = mean / ep
    dlv = -0.5 * (1 - np.exp(log_var) / ep)
    return loss, dmean.astype(mean.dtype), dlv.astype(mean.dtype)


def logistic_normal(w_mean, w_log_var, eps):
    """softmax([mean + exp(lv/2)*eps, 0])  (cl_vae/model.py:146-157; cl_vrnn/model.py:183-191).

    Plain exp / sum like the reference (no max subtraction)."""
    s = w_mean + np.exp(w_log_var / 2) * eps
    s0 = np.concatenate([s, np.zeros(s.shape[:-1] + (1,), s.dtype)], axis=-1)
    e = np.exp(s0)
    return e / e.sum(axis=-1, keepdims=True)


def logistic_normal_bwd(w, dw, w_log_var, eps):
    ds0 = w * (dw - (dw * w).sum(axis=-1, keepdims=True))
    ds = ds0[..., :-1]
    return ds, ds * eps * 0.5 * np.exp(w_log_var / 2)


def categorical_accuracy(onehot, w):
    return float(np.mean(np.argmax(onehot, -1) == np.argmax(w, -1)))


# --------------------------------------------------------------------------- #
# cl_vae  (cl_vae/model.py:130-224)
# --------------------------------------------------------------------------- #
def vae_config(original_dim=88, intermediate_dim=88, latent_dim=2, intermediate_class_dim=88,
               n_classes=2, use_x_prev=False, class_weight=1.0, kl_weight=1.0,
               w_kl_weight=1.0, w_log_var_prior=0.0):
    return dict(D=original_dim, H=intermediate_dim, L=latent_dim, Hc=intermediate_class_dim,
                C=n_classes, use_x_prev=bool(use_x_prev), class_weight=float(class_weight),
                kl_weight=float(kl_weight), w_kl_weight=float(w_kl_weight),
                w_log_var_prior=float(w_log_var_prior))


def vae_param_shapes(cfg):
    """Keras layer/weight order.  intermediate_dim == 0 (cl_vae/model.py:165-167,188): no `h` and no `decoder_h`
    layer -- the latent heads read [x, w] and the output layer reads [w, (history,) z] directly."""
    D, H, L, Hc, C = cfg['D'], cfg['H'], cfg['L'], cfg['Hc'], cfg['C']
    dec_in = C + (D if cfg['use_x_prev'] else 0) + L
    head = [('h_w/kernel', (D, Hc)), ('h_w/bias', (Hc,)),
            ('w_mean/kernel', (Hc, C - 1)), ('w_mean/bias', (C - 1,)),
            ('w_log_var/kernel', (Hc, C - 1)), ('w_log_var/bias', (C - 1,))]
    if H > 0:
        return head + [('h/kernel', (D + C, H)), ('h/bias', (H,)),
                       ('z_mean/kernel', (H, L)), ('z_mean/bias', (L,)),
                       ('z_log_var/kernel', (H, L)), ('z_log_var/bias', (L,)),
                       ('decoder_h/kernel', (dec_in, H)), ('decoder_h/bias', (H,)),
                       ('x_decoded_mean/kernel', (H, D)), ('x_decoded_mean/bias', (D,))]
    return head + [('z_mean/kernel', (D + C, L)), ('z_mean/bias', (L,)),
                   ('z_log_var/kernel', (D + C, L)), ('z_log_var/bias', (L,)),
                   ('x_decoded_mean/kernel', (dec_in, D)), ('x_decoded_mean/bias', (D,))]


def vae_init_params(cfg, seed=0, dtype=np.float64):
    """Keras default initialisers (glorot_uniform kernels, zero biases) [K] A.1."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in vae_param_shapes(cfg):
        p[name] = glorot_uniform(rng, shape, dtype) if len(shape) == 2 else np.zeros(shape, dtype)
    return p


def vae_forward(p, cfg, x, xp, eps_w, eps_z):
    """One forward pass; returns a cache with every intermediate."""
    c = {}
    c['a_hw'] = x @ p['h_w/kernel'] + p['h_w/bias']                       # :141
    c['h_w'] = relu(c['a_hw'])
    c['w_mean'] = c['h_w'] @ p['w_mean/kernel'] + p['w_mean/bias']          # :142
    c['w_log_var'] = c['h_w'] @ p['w_log_var/kernel'] + p['w_log_var/bias']  # :143
    c['w'] = logistic_normal(c['w_mean'], c['w_log_var'], eps_w)            # :146-157
    c['xw'] = np.concatenate([x, c['w']], axis=-1)                          # :160
    if cfg['H'] > 0:
        c['a_h'] = c['xw'] @ p['h/kernel'] + p['h/bias']                    # :162
        c['h'] = relu(c['a_h'])
    else:
        c['h'] = c['xw']                                                    # :165-167: the heads read [x, w] directly
    c['z_mean'] = c['h'] @ p['z_mean/kernel'] + p['z_mean/bias']            # :163
    c['z_log_var'] = c['h'] @ p['z_log_var/kernel'] + p['z_log_var/bias']   # :164
    c['z'] = c['z_mean'] + np.exp(c['z_log_var'] / 2) * eps_z               # :170-174
    if cfg['use_x_prev']:
        c['wz'] = np.concatenate([c['w'], xp, c['z']], axis=-1)             # :177-181 (w, history, z)
    else:
        c['wz'] = np.concatenate([c['w'], c['z']], axis=-1)
    if cfg['H'] > 0:
        c['a_dh'] = c['wz'] @ p['decoder_h/kernel'] + p['decoder_h/bias']   # :184-185
        c['h_dec'] = relu(c['a_dh'])
    else:
        c['h_dec'] = c['wz']                                                # :188: decoder_mean(wz)
    c['logits'] = c['h_dec'] @ p['x_decoded_mean/kernel'] + p['x_decoded_mean/bias']  # :182,186
    c['x_hat'] = sigmoid(c['logits'])
    return c


def vae_loss_and_grads(p, cfg, x, xp, w_true, eps_w, eps_z, need_grads=True, target=None):
    """Forward + the four losses (:190-219) + analytic gradients of the weighted total."""
    B = x.shape[0]
    D, L, C = cfg['D'], cfg['L'], cfg['C']
    c = vae_forward(p, cfg, x, xp, eps_w, eps_z)
    # the decoder output is scored against y[0] of fit(): x itself, or the next frame under --predict_next
    # (cl_vae/train.py:15,66: PianoData(return_y_next=...) and [P.y_train, wtr, wtr, P.y_train])
    vae_b, dlogits = bce_from_logits_keras(c['logits'], x if target is None else target)
    klz_b, dzm_kl, dzlv_kl = kl_gauss(c['z_mean'], c['z_log_var'])
    wrec_b, dw_rec = cce_keras(c['w'], w_true, C - 1)
    klw_b, dwm_kl, dwlv_kl = kl_w_prior(c['w_mean'], c['w_log_var'], cfg['w_log_var_prior'])
    out = dict(vae=vae_b.mean(), kl_z=klz_b.mean(), w_rec=wrec_b.mean(), kl_w=klw_b.mean(),
               acc=categorical_accuracy(w_true, c['w']))
    out['total'] = (out['vae'] + cfg['w_kl_weight'] * out['kl_w'] + cfg['class_weight'] * out['w_rec']
                    + cfg['kl_weight'] * out['kl_z'])                       # :216-218
    out['elbo'] = -(out['vae'] + out['kl_z'] + out['kl_w'] + out['w_rec'])
    out['cache'] = c
    if not need_grads:
        return out
    g = {}
    inv = 1.0 / B
    dlogits = dlogits * inv
    g['x_decoded_mean/kernel'] = c['h_dec'].T @ dlogits
    g['x_decoded_mean/bias'] = dlogits.sum(0)
    if cfg['H'] > 0:
        d = (dlogits @ p['x_decoded_mean/kernel'].T) * (c['a_dh'] > 0)
        g['decoder_h/kernel'] = c['wz'].T @ d
        g['decoder_h/bias'] = d.sum(0)
        dwz = d @ p['decoder_h/kernel'].T
    else:
        dwz = dlogits @ p['x_decoded_mean/kernel'].T
    dw = dwz[:, :C].copy()
    dz = dwz[:, -L:]
    dzm = dz + cfg['kl_weight'] * inv * dzm_kl
    dzlv = dz * eps_z * 0.5 * np.exp(c['z_log_var'] / 2) + cfg['kl_weight'] * inv * dzlv_kl
    g['z_mean/kernel'] = c['h'].T @ dzm
    g['z_mean/bias'] = dzm.sum(0)
    g['z_log_var/kernel'] = c['h'].T @ dzlv
    g['z_log_var/bias'] = dzlv.sum(0)
    if cfg['H'] > 0:
        d = (dzm @ p['z_mean/kernel'].T + dzlv @ p['z_log_var/kernel'].T) * (c['a_h'] > 0)
        g['h/kernel'] = c['xw'].T @ d
        g['h/bias'] = d.sum(0)
        dw += (d @ p['h/kernel'].T)[:, D:]
    else:
        dw += (dzm @ p['z_mean/kernel'].T + dzlv @ p['z_log_var/kernel'].T)[:, D:]
    dw += cfg['class_weight'] * inv * dw_rec
    ds, dlv_from_s = logistic_normal_bwd(c['w'], dw, c['w_log_var'], eps_w)
    dwm = ds + cfg['w_kl_weight'] * inv * dwm_kl
    dwlv = dlv_from_s + cfg['w_kl_weight'] * inv * dwlv_kl
    g['w_mean/kernel'] = c['h_w'].T @ dwm
    g['w_mean/bias'] = dwm.sum(0)
    g['w_log_var/kernel'] = c['h_w'].T @ dwlv
    g['w_log_var/bias'] = dwlv.sum(0)
    d = (dwm @ p['w_mean/kernel'].T + dwlv @ p['w_log_var/kernel'].T) * (c['a_hw'] > 0)
    g['h_w/kernel'] = x.T @ d
    g['h_w/bias'] = d.sum(0)
    out['grads'] = g
    return out


# --------------------------------------------------------------------------- #
# LSTM (Keras 2.0.0 semantics, Appendix A.2)
# --------------------------------------------------------------------------- #
def dropout_masks(u, rate):
    """Keras' input-dropout masks from uniforms u in [0,1): K.dropout(ones, rate) = tf.nn.dropout(ones, keep) =
    ones / keep * floor(keep + u), i.e. 1/keep where u >= rate, else 0  [K]."""
    keep = 1.0 - rate
    return np.floor(keep + u) / keep


def lstm_forward(xs, kernel, rec, bias, h0=None, c0=None, gate_act='hard_sigmoid', in_masks=None):
    """xs [B,T,In] -> hs [B,T,H]; also returns what BPTT needs.

    z = x_t.kernel + bias + h_{t-1}.recurrent ; i,f,o = hs(z_*), g = tanh(z_c)
    c_t = f*c_{t-1} + i*g ; h_t = o*tanh(c_t)    (cl_vrnn/model.py:196-199,225-228 + [K] A.2)
    in_masks [4,B,In] (LSTM(dropout=p) in the training phase, cl_vrnn/model.py:198,227; [K] Keras 2.0.0 recurrent.py,
    implementation 0: `_time_distributed_dense(x, kernel_g, bias_g, dropout)` draws ONE mask per gate and sample, constant
    over the time steps, and multiplies the inputs of that gate's projection with it): z_g = (x_t * m_g).kernel_g + ...
    """
    B, T, _ = xs.shape
    H = rec.shape[0]
    act = hard_sigmoid if gate_act == 'hard_sigmoid' else sigmoid
    h = np.zeros((B, H), xs.dtype) if h0 is None else h0
    c = np.zeros((B, H), xs.dtype) if c0 is None else c0
    if in_masks is None:
        xproj = xs @ kernel + bias                   # implementation=0 precompute [K]
    else:
        xproj = np.concatenate([(xs * in_masks[g][:, None, :]) @ kernel[:, g * H:(g + 1) * H] for g in range(4)], axis=-1) + bias
    Z = np.empty((B, T, 4 * H), xs.dtype)
    Cs = np.empty((B, T, H), xs.dtype)
    Hs = np.empty((B, T, H), xs.dtype)
    for t in range(T):
        z = xproj[:, t] + h @ rec
        i = act(z[:, :H]); f = act(z[:, H:2 * H]); g = np.tanh(z[:, 2 * H:3 * H]); o = act(z[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        Z[:, t] = z; Cs[:, t] = c; Hs[:, t] = h
    return Hs, dict(Z=Z, C=Cs, H=Hs, xs=xs, h0=h0, c0=c0, gate_act=gate_act, in_masks=in_masks)


def lstm_backward(dHs, cache, kernel, rec):
    """BPTT given dL/dh_t for every t.  Returns (dxs, dkernel, drec, dbias)."""
    Z, Cs, Hs, xs = cache['Z'], cache['C'], cache['H'], cache['xs']
    B, T, H = Hs.shape
    if cache['gate_act'] == 'hard_sigmoid':
        act, dact = hard_sigmoid, hard_sigmoid_grad
    else:
        act = sigmoid
        dact = lambda z: sigmoid(z) * (1 - sigmoid(z))
    dZ = np.empty_like(Z)
    dh_rec = np.zeros((B, H), Z.dtype)
    dc = np.zeros((B, H), Z.dtype)
    for t in range(T - 1, -1, -1):
        z = Z[:, t]
        i = act(z[:, :H]); f = act(z[:, H:2 * H]); g = np.tanh(z[:, 2 * H:3 * H]); o = act(z[:, 3 * H:])
        c_prev = Cs[:, t - 1] if t > 0 else (np.zeros((B, H), Z.dtype) if cache['c0'] is None else cache['c0'])
        tc = np.tanh(Cs[:, t])
        dh = dHs[:, t] + dh_rec
        do = dh * tc
        dc = dc + dh * o * (1 - tc * tc)
        dzt = np.concatenate([dc * g * dact(z[:, :H]), dc * c_prev * dact(z[:, H:2 * H]),
                              dc * i * (1 - g * g), do * dact(z[:, 3 * H:])], axis=-1)
        dZ[:, t] = dzt
        dc = dc * f
        dh_rec = dzt @ rec.T
    h_prev = np.concatenate([np.zeros((B, 1, H), Z.dtype) if cache['h0'] is None else cache['h0'][:, None],
                             Hs[:, :-1]], axis=1)
    dZ2 = dZ.reshape(B * T, 4 * H)
    drec = h_prev.reshape(B * T, H).T @ dZ2
    dbias = dZ2.sum(0)
    masks = cache.get('in_masks')
    if masks is None:
        dkernel = xs.reshape(B * T, -1).T @ dZ2
        dxs = (dZ2 @ kernel.T).reshape(B, T, -1)
    else:                                            # per gate: the masked inputs' products
        dkernel = np.concatenate([(xs * masks[g][:, None, :]).reshape(B * T, -1).T @ dZ2[:, g * H:(g + 1) * H]
                                  for g in range(4)], axis=-1)
        dxs = sum((dZ2[:, g * H:(g + 1) * H] @ kernel[:, g * H:(g + 1) * H].T).reshape(B, T, -1) * masks[g][:, None, :]
                  for g in range(4))
    return dxs, dkernel, drec, dbias, dZ


# --------------------------------------------------------------------------- #
# cl_vrnn  (cl_vrnn/model.py:164-267)
# --------------------------------------------------------------------------- #
def vrnn_config(original_dim=88, intermediate_dim=88, latent_dim=2, seq_length=16, n_classes=10,
                use_x_prev=True, class_weight=1.0, kl_weight=1.0, w_kl_weight=1.0,
                w_log_var_prior=0.0, gate_act='hard_sigmoid'):
    return dict(D=original_dim, H=intermediate_dim, L=latent_dim, T=seq_length, C=n_classes,
                use_x_prev=bool(use_x_prev), class_weight=float(class_weight),
                kl_weight=float(kl_weight), w_kl_weight=float(w_kl_weight),
                w_log_var_prior=float(w_log_var_prior), gate_act=gate_act)


def vrnn_param_shapes(cfg):
    D, H, L, T, C = cfg['D'], cfg['H'], cfg['L'], cfg['T'], cfg['C']
    dec_in = (D if cfg['use_x_prev'] else 0) + L + C
    return [('hW/kernel', (T * D, D)), ('hW/bias', (D,)),
            ('Wargs/kernel', (D, 2 * (C - 1))), ('Wargs/bias', (2 * (C - 1),)),
            ('encoder_h/kernel', (D + C, 4 * H)), ('encoder_h/recurrent_kernel', (H, 4 * H)),
            ('encoder_h/bias', (4 * H,)),
            ('Z_mean/kernel', (H, L)), ('Z_mean/bias', (L,)),
            ('Z_log_var/kernel', (H, L)), ('Z_log_var/bias', (L,)),
            ('decoder_h/kernel', (dec_in, 4 * H)), ('decoder_h/recurrent_kernel', (H, 4 * H)),
            ('decoder_h/bias', (4 * H,)),
            ('X_decoded_mean/kernel', (H, D)), ('X_decoded_mean/bias', (D,))]


def vrnn_init_params(cfg, seed=0, dtype=np.float64):
    """Initialisers per [K] A.1 and cl_vrnn/model.py:200-207,229-233 (N(0,0.1) heads)."""
    rng = np.random.default_rng(seed)
    H = cfg['H']
    p = {}
    for name, shape in vrnn_param_shapes(cfg):
        layer, wname = name.split('/')
        if wname == 'bias':
            b = np.zeros(shape, dtype)
            if layer in ('encoder_h', 'decoder_h'):
                b[H:2 * H] = 1.0                                    # unit_forget_bias
            p[name] = b
        elif wname == 'recurrent_kernel':
            p[name] = orthogonal(rng, shape, dtype)
        elif layer in ('Z_mean', 'Z_log_var', 'X_decoded_mean'):
            p[name] = (0.1 * rng.standard_normal(shape)).astype(dtype)
        else:
            p[name] = glorot_uniform(rng, shape, dtype)
    return p


def vrnn_forward(p, cfg, X, Xp, eps_W, eps_Z, masks=None):
    """X, Xp [B,T,D]; eps_W [B,C-1]; eps_Z [B,T,L]  (SURVEY.md 3.3).
    masks = (enc [4,B,D+C], dec [4,B,(D)+L+C]): the input-dropout masks of the two LSTMs (training phase of
    get_model(dropout=p), cl_vrnn/model.py:164,198,227); None: no dropout (the reference's scripts, and every inference)."""
    B, T, D = X.shape
    C, L = cfg['C'], cfg['L']
    c = {}
    c['Xflat'] = X.reshape(B, T * D)                                        # Flatten :174
    c['a_hW'] = c['Xflat'] @ p['hW/kernel'] + p['hW/bias']
    c['hW'] = relu(c['a_hW'])
    c['Wargs'] = c['hW'] @ p['Wargs/kernel'] + p['Wargs/bias']              # :175
    c['W_mean'] = c['Wargs'][:, :C - 1]                                     # :176-181
    c['W_log_var'] = c['Wargs'][:, C - 1:]
    c['W'] = logistic_normal(c['W_mean'], c['W_log_var'], eps_W)            # :183-191
    Wrep = np.repeat(c['W'][:, None, :], T, axis=1)
    c['XW'] = np.concatenate([X, Wrep], axis=-1)                            # :193
    c['enc_h'], c['enc_cache'] = lstm_forward(c['XW'], p['encoder_h/kernel'],
                                              p['encoder_h/recurrent_kernel'], p['encoder_h/bias'],
                                              gate_act=cfg['gate_act'], in_masks=None if masks is None else masks[0])     # :196-199
    c['Z_mean'] = c['enc_h'] @ p['Z_mean/kernel'] + p['Z_mean/bias']        # :200-209
    c['Z_log_var'] = c['enc_h'] @ p['Z_log_var/kernel'] + p['Z_log_var/bias']
    c['Z'] = c['Z_mean'] + np.exp(c['Z_log_var'] / 2) * eps_Z               # :212-216
    if cfg['use_x_prev']:
        c['XpZ'] = np.concatenate([Xp, c['Z'], Wrep], axis=-1)              # :218-222 (history, z, w)
    else:
        c['XpZ'] = np.concatenate([c['Z'], Wrep], axis=-1)
    c['dec_h'], c['dec_cache'] = lstm_forward(c['XpZ'], p['decoder_h/kernel'],
                                              p['decoder_h/recurrent_kernel'], p['decoder_h/bias'],
                                              gate_act=cfg['gate_act'], in_masks=None if masks is None else masks[1])     # :225-228
    c['logits'] = c['dec_h'] @ p['X_decoded_mean/kernel'] + p['X_decoded_mean/bias']  # :229-234
    c['X_hat'] = sigmoid(c['logits'])
    return c


def vrnn_loss_and_grads(p, cfg, X, Xp, w_true, eps_W, eps_Z, need_grads=True, target=None, masks=None):
    B, T, D = X.shape
    C, L, H = cfg['C'], cfg['L'], cfg['H']
    c = vrnn_forward(p, cfg, X, Xp, eps_W, eps_Z, masks=masks)
    # scored against y[0] of fit() (cl_vrnn/train.py:15,59-66): X, or the next frames under --predict_next
    vae_bt, dlogits = bce_from_logits_keras(c['logits'], X if target is None else target)                # [B,T]
    klz_bt, dzm_kl, dzlv_kl = kl_gauss(c['Z_mean'], c['Z_log_var'])
    wrec_b, dw_rec = cce_keras(c['W'], w_true, C - 1)
    klw_b, dwm_kl, dwlv_kl = kl_w_prior(c['W_mean'], c['W_log_var'], cfg['w_log_var_prior'])
    out = dict(vae=vae_bt.mean(), kl_z=klz_bt.mean(), w_rec=wrec_b.mean(), kl_w=klw_b.mean(),
               acc=categorical_accuracy(w_true, c['W']))
    out['total'] = (out['vae'] + cfg['w_kl_weight'] * out['kl_w'] + cfg['class_weight'] * out['w_rec']
                    + cfg['kl_weight'] * out['kl_z'])                       # :261-263
    out['elbo'] = -(out['vae'] + out['kl_z'] + out['kl_w'] + out['w_rec'])
    out['cache'] = c
    if not need_grads:
        return out
    g = {}
    inv_bt = 1.0 / (B * T)
    inv_b = 1.0 / B
    dlogits = (dlogits * inv_bt).reshape(B * T, D)
    dech = c['dec_h'].reshape(B * T, H)
    g['X_decoded_mean/kernel'] = dech.T @ dlogits
    g['X_decoded_mean/bias'] = dlogits.sum(0)
    d_dec_h = (dlogits @ p['X_decoded_mean/kernel'].T).reshape(B, T, H)
    dXpZ, g['decoder_h/kernel'], g['decoder_h/recurrent_kernel'], g['decoder_h/bias'], _ = \
        lstm_backward(d_dec_h, c['dec_cache'], p['decoder_h/kernel'], p['decoder_h/recurrent_kernel'])
    off = D if cfg['use_x_prev'] else 0
    dZ = dXpZ[:, :, off:off + L]
    dW = dXpZ[:, :, off + L:].sum(axis=1)                                  # RepeatVector
    dZm = dZ + cfg['kl_weight'] * inv_bt * dzm_kl
    dZlv = dZ * eps_Z * 0.5 * np.exp(c['Z_log_var'] / 2) + cfg['kl_weight'] * inv_bt * dzlv_kl
    ench = c['enc_h'].reshape(B * T, H)
    g['Z_mean/kernel'] = ench.T @ dZm.reshape(B * T, L)
    g['Z_mean/bias'] = dZm.reshape(B * T, L).sum(0)
    g['Z_log_var/kernel'] = ench.T @ dZlv.reshape(B * T, L)
    g['Z_log_var/bias'] = dZlv.reshape(B * T, L).sum(0)
    d_enc_h = dZm @ p['Z_mean/kernel'].T + dZlv @ p['Z_log_var/kernel'].T
    dXW, g['encoder_h/kernel'], g['encoder_h/recurrent_kernel'], g['encoder_h/bias'], _ = \
        lstm_backward(d_enc_h, c['enc_cache'], p['encoder_h/kernel'], p['encoder_h/recurrent_kernel'])
    dW = dW + dXW[:, :, D:].sum(axis=1)
    dW = dW + cfg['class_weight'] * inv_b * dw_rec
    ds, dlv_from_s = logistic_normal_bwd(c['W'], dW, c['W_log_var'], eps_W)
    dWm = ds + cfg['w_kl_weight'] * inv_b * dwm_kl
    dWlv = dlv_from_s + cfg['w_kl_weight'] * inv_b * dwlv_kl
    dWargs = np.concatenate([dWm, dWlv], axis=-1)
    g['Wargs/kernel'] = c['hW'].T @ dWargs
    g['Wargs/bias'] = dWargs.sum(0)
    d = (dWargs @ p['Wargs/kernel'].T) * (c['a_hW'] > 0)
    g['hW/kernel'] = c['Xflat'].T @ d
    g['hW/bias'] = d.sum(0)
    out['grads'] = g
    return out


# --------------------------------------------------------------------------- #
# Adam with weight normalisation (utils/weightnorm.py:75-178; [K] A.5)
# --------------------------------------------------------------------------- #
def adam_wn_init(params, weightnorm=True):
    st = {'t': 0, 'weightnorm': bool(weightnorm), 'm': {}, 'v': {}, 'mg': {}, 'vg': {}, 's': {}}
    for k, p in params.items():
        st['m'][k] = np.zeros_like(p)
        st['v'][k] = np.zeros_like(p)
        if weightnorm and p.ndim > 1:
            st['mg'][k] = np.zeros(p.shape[-1], p.dtype)
            st['vg'][k] = np.zeros(p.shape[-1], p.dtype)
            st['s'][k] = np.ones(p.shape[-1], p.dtype)              # V_scaler init (:153)
    return st


def adam_wn_step(params, grads, st, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """In-place update of params and st.  weightnorm=False gives plain Keras Adam."""
    st['t'] += 1
    t = st['t']                                                     # t = iterations + 1 (:84)
    lr_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)            # :85
    for k, p in params.items():
        g = grads[k]
        m, v = st['m'][k], st['v'][k]
        if st['weightnorm'] and p.ndim > 1:                         # :96-127
            s = st['s'][k]
            V = p / s                                               # :157
            ax = tuple(range(p.ndim - 1))
            Vn = np.sqrt((V * V).sum(axis=ax))                      # :160
            gp = s * Vn                                             # :161
            grad_g = (g * V).sum(axis=ax) / Vn                      # :164
            grad_V = s * (g - (grad_g / Vn) * V)                    # :165-166
            mg, vg = st['mg'][k], st['vg'][k]
            mg[...] = b1 * mg + (1 - b1) * grad_g                   # :107
            vg[...] = b2 * vg + (1 - b2) * grad_g ** 2              # :108
            new_g = gp - lr_t * mg / (np.sqrt(vg) + eps)            # :109
            m[...] = b1 * m + (1 - b1) * grad_V                     # :114
            v[...] = b2 * v + (1 - b2) * grad_V ** 2                # :115
            new_V = V - lr_t * m / (np.sqrt(v) + eps)               # :116
            nVn = np.sqrt((new_V * new_V).sum(axis=ax))             # :174
            new_s = new_g / nVn                                     # :175
            p[...] = new_s * new_V                                  # :176
            s[...] = new_s                                          # :178
        else:                                                       # :129-142
            m[...] = b1 * m + (1 - b1) * g
            v[...] = b2 * v + (1 - b2) * g ** 2
            p[...] = p - lr_t * m / (np.sqrt(v) + eps)
    return params


def rmsprop_step(params, grads, acc, lr=1e-3, rho=0.9, eps=1e-8):
    """Keras 2.0.0 RMSprop (optimizers.py, what the string 'rmsprop' of cl_vae/train.py:83 selects [K]):
    a <- rho a + (1 - rho) g^2 ; p <- p - lr g / (sqrt(a) + eps).  `acc`: dict of accumulators (zeros at the start)."""
    for k, p in params.items():
        a = acc[k]
        a[...] = rho * a + (1.0 - rho) * grads[k] ** 2
        p[...] = p - lr * grads[k] / (np.sqrt(a) + eps)
    return params
