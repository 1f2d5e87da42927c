"""Keras 2.0.0 initialisers for the layers of cl_vae / cl_vrnn (host side, numpy).

Only the distributions need to match the reference, not TF's RNG streams:
Dense kernels glorot_uniform, biases zero; the cl_vrnn heads RandomNormal(0, 0.1)
(cl_vrnn/model.py:200-207,229-233); LSTM kernel glorot_uniform over (in, 4H),
recurrent_kernel orthogonal, bias zero with unit_forget_bias (bias[H:2H] = 1).
"""
import numpy as np

_NORMAL_HEADS = ('Z_mean', 'Z_log_var', 'X_decoded_mean')   # cl_vrnn only


def glorot_uniform(rng, shape):
    lim = np.sqrt(6.0 / (shape[0] + shape[1]))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def orthogonal(rng, shape):
    a = rng.standard_normal(shape)
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    return (u if u.shape == tuple(shape) else vt).astype(np.float32)


def init_weights(shapes, cfg, seed=None):
    """{name: array} for a list of ('<layer>/<weight>', shape) in Keras layer order."""
    rng = np.random.default_rng(seed)
    is_vrnn = 'T' in cfg
    H = cfg['H']
    out = {}
    for name, shape in shapes:
        layer, wname = name.split('/')
        if wname == 'bias':
            b = np.zeros(shape, np.float32)
            if is_vrnn and layer in ('encoder_h', 'decoder_h'):
                b[H:2 * H] = 1.0
            out[name] = b
        elif wname == 'recurrent_kernel':
            out[name] = orthogonal(rng, shape)
        elif is_vrnn and layer in _NORMAL_HEADS:
            out[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = glorot_uniform(rng, shape)
    return out
