#!/usr/bin/env python
"""G6: outputs of the reference's OWN lines for the two slices of the arithmetic that need no Keras layer semantics.
Runs ONLY in the build container (reads /root/reference); writes tests/golden/g6_reference_lines.npz (arrays only).

(a) utils/weightnorm.py:75-178 -- `AdamWithWeightnorm.get_updates` and its two helpers -- exec'd as they stand, with
    `keras.optimizers.Adam`, `keras.backend` and `tensorflow` replaced by a numpy namespace that supplies exactly the
    calls those lines make:
        K.update_add / K.update (recorded, applied after get_updates returns: every new value is computed from the OLD
        values, which is what a TF-1 `sess.run(updates)` of read-then-assign ops does), K.zeros / K.ones (optimizer state:
        created at the first call, the same array at every later call, in call order), K.get_variable_shape, K.sqrt,
        K.pow, K.square; tf.reshape, tf.sqrt, tf.square, tf.reduce_sum(x, axes).
    `get_updates` builds a graph once in Keras; here it is called once per step on the current values, which evaluates
    the same expressions.  What this pins: the (V, g) reparametrisation, `t = iterations + 1`, the bias correction, which
    tensors are weight-normalised (ndim > 1), the axis of the norm, the order of the updates, the state's initial values.
(b) the loss closures cl_vae/model.py:193-196 (kl_loss), :202-206 (w_kl_loss) and cl_vrnn/model.py:236-239, :247-252,
    dedented and exec'd with K.{sum, square, exp} = numpy's.  `vae_loss` / `w_rec_loss` (:190-191, :198-199) call
    keras.losses, which is not in the checkout: they stay [K]-recalled (oracle/clvae_oracle.py).
Layer semantics (Dense, LSTM, TimeDistributed, the initialisers) stay recalled as well; see DESIGN.md section 2.
"""
import os
import sys
import textwrap

import numpy as np

REF = '/root/reference/code'
HERE = os.path.dirname(os.path.abspath(__file__))


class Var(np.ndarray):
    """an optimizer / model variable: a numpy array that can be a dictionary key (`p in constraints`)"""
    def __hash__(self):
        return id(self)


def var(a):
    return np.array(a, dtype=np.float64).view(Var)


class KShim:
    def __init__(self):
        self.state, self.cursor, self.pending = [], 0, []

    def begin(self):
        self.cursor, self.pending = 0, []

    def _state(self, shape, fill):
        if self.cursor == len(self.state):
            self.state.append(var(np.full(shape, fill)))
        v = self.state[self.cursor]
        assert v.shape == tuple(shape)
        self.cursor += 1
        return v

    def zeros(self, shape):
        return self._state(tuple(shape), 0.0)

    def ones(self, shape):
        return self._state(tuple(shape), 1.0)

    def get_variable_shape(self, p):
        return tuple(np.shape(p))

    def update(self, x, new):
        self.pending.append((x, np.array(new, dtype=np.float64)))
        return ('update', id(x))

    def update_add(self, x, inc):
        self.pending.append((x, inc))
        return ('update_add', id(x))

    sqrt = staticmethod(np.sqrt)
    square = staticmethod(np.square)
    pow = staticmethod(np.power)
    exp = staticmethod(np.exp)

    @staticmethod
    def sum(x, axis=None):
        return np.sum(x, axis=axis)


class TfShim:
    reshape = staticmethod(lambda x, shape: np.reshape(x, shape))
    sqrt = staticmethod(np.sqrt)
    square = staticmethod(np.square)

    @staticmethod
    def reduce_sum(x, axes):
        return np.sum(x, axis=tuple(axes))


class Iterations:
    """`self.iterations`: supports `+ 1` and `* decay`, advanced by K.update_add"""
    def __init__(self):
        self.n = 0

    def __add__(self, k):
        return self.n + k

    __radd__ = __add__

    def __mul__(self, k):
        return self.n * k

    __rmul__ = __mul__


class FakeAdam:
    """what utils/weightnorm.py:75-143 reads of keras.optimizers.Adam (Keras 2.0: lr 0.001, beta_1 0.9, beta_2 0.999,
    epsilon 1e-8, decay 0; model_utils.py:54 passes lr / beta / epsilon / decay exactly so)"""
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.):
        self.iterations = Iterations()
        self.lr, self.beta_1, self.beta_2, self.epsilon, self.decay = lr, beta_1, beta_2, epsilon, decay
        self.initial_decay = decay
        self._grads = None

    def get_gradients(self, loss, params):
        return self._grads


def exec_lines(path, first, last, ns, dedent=False):
    with open(path) as f:
        lines = f.readlines()
    src = ''.join(lines[first - 1:last])
    if dedent:
        src = textwrap.dedent(src)
    exec(compile(src, path, 'exec'), ns)
    return ns


def optimizer_trace():
    K, tf = KShim(), TfShim()
    ns = exec_lines(os.path.join(REF, 'utils/weightnorm.py'), 75, 178, dict(K=K, tf=tf, Adam=FakeAdam, np=np))
    opt = ns['AdamWithWeightnorm'](lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-8, decay=0.)
    rng = np.random.default_rng(606)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    shapes = [('tall/kernel', (160, 8)),      # more than 144 rows: the tensor the HIP optimizer's two-launch form applies to
              ('dense/kernel', (13, 5)), ('dense/bias', (5,)), ('lstm/kernel', (7, 12)), ('head/kernel', (6, 1)),
              ('head/bias', (1,))]
    params = [var(f32(rng.standard_normal(s) * 0.3)) for _, s in shapes]
    out = {}
    for (name, _), p in zip(shapes, params):
        out['opt/p0/' + name] = np.array(p)
    nsteps = 4
    for step in range(nsteps):
        # gradients of very different sizes per tensor, some entries exactly zero (rows of notes that never sound)
        grads = []
        for i, (_, s) in enumerate(shapes):
            g = f32(rng.standard_normal(s) * 10.0 ** (i - 4))
            if len(s) > 1:
                g[0] = 0.0
            grads.append(g)
            out['opt/g%d/%s' % (step, shapes[i][0])] = g
        opt._grads = grads
        K.begin()
        opt.get_updates(params, {}, None)
        for x, new in K.pending:                       # every new value was formed from the old ones: assign now
            if isinstance(x, Iterations):
                x.n += new
            else:
                x[...] = new
        for (name, _), p in zip(shapes, params):
            out['opt/p%d/%s' % (step + 1, name)] = np.array(p)
    assert opt.iterations.n == nsteps
    # the optimizer state in creation order: ms (6), vs (6), then per weight-normalised tensor V_scaler, m_g, v_g
    out['opt/n_state'] = np.array(len(K.state))
    for i, s in enumerate(K.state):
        out['opt/state/%02d' % i] = np.array(s)
    out['opt/names'] = np.array([n for n, _ in shapes])
    return out


def loss_closures():
    out = {}
    rng = np.random.default_rng(607)
    K = KShim()
    # ---- cl_vae/model.py:193-196 and :202-206 (bodies of get_model: dedented) -------------------------------------------
    B, L, C1 = 9, 4, 1
    z_args = rng.standard_normal((B, 2 * L))
    w_mean, w_log_var = rng.standard_normal((B, C1)), rng.standard_normal((B, C1)) * 0.7
    for prior in (0.0, 0.5, -1.0):
        ns = dict(K=K, latent_dim=L, w_log_var_prior=prior, w_mean=w_mean, w_log_var=w_log_var)
        exec_lines(os.path.join(REF, 'cl_vae/model.py'), 193, 196, ns, dedent=True)
        exec_lines(os.path.join(REF, 'cl_vae/model.py'), 202, 206, ns, dedent=True)
        out['loss/vae/kl_z'] = ns['kl_loss'](None, z_args)
        out['loss/vae/kl_w/prior%g' % prior] = ns['w_kl_loss'](None, None)
    out.update({'loss/vae/z_args': z_args, 'loss/vae/w_mean': w_mean, 'loss/vae/w_log_var': w_log_var})
    # ---- cl_vrnn/model.py:236-239 and :247-252 ----------------------------------------------------------------------------
    B, T, L, C1 = 5, 7, 3, 9
    Z_args = rng.standard_normal((B, T, 2 * L))
    W_mean, W_log_var = rng.standard_normal((B, C1)), rng.standard_normal((B, C1)) * 0.7
    for prior in (0.0, 0.5, -1.0):
        ns = dict(K=K, latent_dim=L, w_log_var_prior=prior, W_mean=W_mean, W_log_var=W_log_var)
        exec_lines(os.path.join(REF, 'cl_vrnn/model.py'), 236, 239, ns, dedent=True)
        exec_lines(os.path.join(REF, 'cl_vrnn/model.py'), 247, 252, ns, dedent=True)
        out['loss/vrnn/kl_z'] = ns['kl_loss'](None, Z_args)
        out['loss/vrnn/kl_w/prior%g' % prior] = ns['w_kl_loss'](None, None)
    out.update({'loss/vrnn/Z_args': Z_args, 'loss/vrnn/W_mean': W_mean, 'loss/vrnn/W_log_var': W_log_var})
    return out


if __name__ == '__main__':
    out = {}
    out.update(optimizer_trace())
    out.update(loss_closures())
    path = os.path.join(HERE, 'g6_reference_lines.npz')
    np.savez_compressed(path, **out)
    print('G6:', len(out), 'arrays,', os.path.getsize(path) // 1024, 'KiB')
