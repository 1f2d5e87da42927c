#!/usr/bin/env python
"""Regenerates tests/golden/*.npz.  Runs ONLY in the build container (needs /root/reference).

G1  PianoData tensors: the reference's own utils/pianoroll.py is imported (Python-2 shims:
    cPickle -> pickle re-opened in binary mode, xrange -> range) and run on the real JSB pickles.
G2  sample_x / sample_w / sample_z / sample_w_discrete: the numpy-only line ranges of the two
    model.py files are exec'd under np.random.seed(k).
G3  generate_sample traces of both models driven by deterministic stub models.
G4  the fp64 oracle's own outputs for fixed weights / inputs / eps (pins the HIP kernels against
    committed numbers rather than against whatever the oracle computes at test time).
Only arrays are stored -- no reference source text.
"""
import builtins
import hashlib
import io
import os
import pickle
import sys
import types

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def import_reference_pianoroll():
    shim = types.ModuleType('cPickle')

    def load(f):
        with open(f.name, 'rb') as g:
            return pickle.load(g, encoding='latin1')
    shim.load = load
    sys.modules['cPickle'] = shim
    builtins.xrange = range
    sys.path.insert(0, os.path.join(REF, 'code'))
    import importlib
    return importlib.import_module('utils.pianoroll')


def exec_slice(path, first, last, extra=None):
    """exec lines [first, last] (1-based, inclusive) of a reference file in a fresh namespace."""
    with open(path) as f:
        lines = f.readlines()
    ns = {'np': np, 'xrange': range}
    ns.update(extra or {})
    exec(compile(''.join(lines[first - 1:last]), path, 'exec'), ns)
    return ns


def checksum(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], dtype=np.uint64)[0]


def g1():
    ref = import_reference_pianoroll()
    out = {}
    cases = [('Cs', 100, 1, dict(return_y_next=True, squeeze_x=True, squeeze_y=True)),
             ('Cs', 512, 1, dict(return_y_next=True, squeeze_x=True, squeeze_y=True)),
             ('all', 200, 16, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
             ('all', 256, 32, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
             ('all', 256, 64, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
             ('all', 256, 128, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
             ('Cs', 1, 32, dict(return_y_next=False, squeeze_x=False, squeeze_y=False)),
             ('all', None, 1, dict())]
    for name, bs, T, kw in cases:
        f = os.path.join(REF, 'data', 'input', 'JSB Chorales_%s.pickle' % name)
        P = ref.PianoData(f, batch_size=bs, seq_length=T, step_length=1, **kw)
        tag = '%s_b%s_t%d' % (name, bs, T)
        for split in ('train', 'valid', 'test'):
            for xy in ('x', 'y'):
                a = getattr(P, '%s_%s' % (xy, split))
                out['%s/%s_%s/shape' % (tag, xy, split)] = np.array(a.shape)
                out['%s/%s_%s/sum' % (tag, xy, split)] = np.array(a.sum())
                out['%s/%s_%s/sha' % (tag, xy, split)] = np.array(checksum(a.astype(np.uint8)))
                if a.shape[0]:
                    out['%s/%s_%s/head' % (tag, xy, split)] = a[:4].astype(np.uint8)
                    out['%s/%s_%s/tail' % (tag, xy, split)] = a[-4:].astype(np.uint8)
            out['%s/%s_song_keys' % (tag, split)] = getattr(P, '%s_song_keys' % split).astype(np.int16)
            out['%s/%s_song_inds' % (tag, split)] = getattr(P, '%s_song_inds' % split).astype(np.int16)
            out['%s/%s_song_modes' % (tag, split)] = getattr(P, '%s_song_modes' % split).astype(np.uint8)
        keys = sorted(P.key_map, key=lambda k: P.key_map[k])
        out['%s/key_map' % tag] = np.array([str(k) for k in keys])
    # helper functions on a synthetic song that needs both octave shifts
    song = [(20, 30), (50, 60, 108), (109,), (21,)]
    out['helpers/song_to_pianoroll_low'] = ref.song_to_pianoroll([(20, 30), (50,)]).astype(np.uint8)
    out['helpers/song_to_pianoroll_high'] = ref.song_to_pianoroll([(50, 109), (60,)]).astype(np.uint8)
    out['helpers/sliding_window'] = ref.sliding_window(np.arange(7 * 88).reshape(7, 88) % 5, 3, 2).astype(np.int16)
    np.savez_compressed(os.path.join(HERE, 'g1_pianodata.npz'), **out)
    print('G1:', len(out), 'arrays')


class StubModel:
    """Deterministic stand-in for a Keras sub-model: predict() is a fixed smooth function of its inputs."""

    def __init__(self, kind, dims, log):
        self.kind, self.dims, self.log = kind, dims, log
        self.n_reset = 0
        self.state = 0.0

    def reset_states(self):
        self.n_reset += 1
        self.state = 0.0
        self.log.append((self.kind, 'reset'))

    def predict(self, x):
        xs = x if isinstance(x, list) else [x]
        feat = sum(float(np.sum(np.asarray(a) * (1 + np.arange(np.asarray(a).size).reshape(np.asarray(a).shape) % 7)))
                   for a in xs)
        self.state = 0.5 * self.state + 0.01 * feat          # "stateful"
        self.log.append((self.kind, [tuple(np.asarray(a).shape) for a in xs]))
        base = np.sin(self.state + np.arange(max(self.dims)))
        if self.kind == 'w_enc':
            C1 = self.dims[0]
            return [base[:C1][None, :] * 0.5, base[:C1][None, :] * 0.1 - 1.0]
        if self.kind == 'z_enc':
            L, lead = self.dims
            shp = (1,) * lead + (L,)
            return [(base[:L] * 0.3).reshape(shp), (base[:L] * 0.1 - 0.5).reshape(shp)]
        D, lead = self.dims
        return (1 / (1 + np.exp(-3 * base[:D]))).reshape((1,) * lead + (D,))


def g2_g3():
    out = {}
    vae = exec_slice(os.path.join(REF, 'code/cl_vae/model.py'), 9, 74)
    vrnn = exec_slice(os.path.join(REF, 'code/cl_vrnn/model.py'), 9, 96)
    mu = np.linspace(-1, 1, 9)[None, :]
    lv = np.linspace(-2, 0.5, 9)[None, :]
    p = np.linspace(0.01, 0.99, 88)[None, :]
    for name, ns in (('vae', vae), ('vrnn', vrnn)):
        for seed in (0, 7):
            np.random.seed(seed); out['g2/%s/sample_x/%d' % (name, seed)] = ns['sample_x'](p)
            np.random.seed(seed); out['g2/%s/sample_w/%d' % (name, seed)] = ns['sample_w']((mu, lv))
            np.random.seed(seed); out['g2/%s/sample_w_nonoise/%d' % (name, seed)] = ns['sample_w']((mu, lv), add_noise=False)
            np.random.seed(seed); out['g2/%s/sample_w_n3/%d' % (name, seed)] = ns['sample_w']((mu, lv), nsamps=3)
            np.random.seed(seed); out['g2/%s/sample_w_nrm/%d' % (name, seed)] = ns['sample_w']((mu, lv), nrm_samp=True)
            np.random.seed(seed); out['g2/%s/sample_z/%d' % (name, seed)] = ns['sample_z']((mu[:, :4], lv[:, :4]))
            np.random.seed(seed); out['g2/%s/sample_z_n2/%d' % (name, seed)] = ns['sample_z']((mu[:, :4], lv[:, :4]), nsamps=2)
        if name == 'vrnn':
            np.random.seed(3); out['g2/vrnn/sample_w_discrete/3'] = ns['sample_w_discrete'](np.array([0.1, 0.2, 0.3, 0.4]))
    out['g2/mu'], out['g2/lv'], out['g2/p'] = mu, lv, p

    # G3 cl_vae generate_sample
    for tag, kw in (('infer_w', dict(w_val=None, use_x_prev=True)),
                    ('given_w', dict(w_val=np.array([[0.25, 0.75]]), use_x_prev=False)),
                    ('z_prior', dict(w_val=None, use_z_prior=True, use_x_prev=True, w_sample=True))):
        log = []
        dec, wenc, zenc = StubModel('dec', (88, 2), log), StubModel('w_enc', (1,), log), StubModel('z_enc', (4, 2), log)
        x_seed = (np.arange(88) % 11 == 0).astype(float)
        np.random.seed(11)
        Xs = vae['generate_sample'](dec, wenc, zenc, x_seed, 6, **kw)
        out['g3/vae/%s/Xs' % tag] = Xs
        out['g3/vae/%s/ncalls' % tag] = np.array([sum(1 for l in log if l[0] == k) for k in ('dec', 'w_enc', 'z_enc')])
    # G3 cl_vrnn generate_sample (seeded teacher forcing, w inference by chunks, discrete w)
    T = 4
    for tag, kw in (('infer_w', dict(w_val=None, seq_length=T)),
                    ('discrete_w', dict(w_val=None, seq_length=T, w_discrete=True)),
                    ('given_w', dict(w_val=np.eye(10)[3][None, :], seq_length=T)),
                    ('no_x_prev', dict(w_val=np.eye(10)[1][None, :], seq_length=T))):
        log = []
        dec, wenc, zenc = StubModel('dec', (88, 3), log), StubModel('w_enc', (9,), log), StubModel('z_enc', (2, 3), log)
        rng = np.random.RandomState(5)
        x_seed = (rng.rand(96, 88) < 0.05).astype(float)       # 96 >= 88: the reference slices time by shape[1] (B5)
        np.random.seed(13)
        Xs = vrnn['generate_sample'](dec, wenc, zenc, x_seed, 5, tag != 'no_x_prev', **kw)
        out['g3/vrnn/%s/Xs' % tag] = Xs
        out['g3/vrnn/%s/ncalls' % tag] = np.array([sum(1 for l in log if l[0] == k and l[1] != 'reset')
                                                   for k in ('dec', 'w_enc', 'z_enc')])
        out['g3/vrnn/%s/nreset' % tag] = np.array([dec.n_reset, wenc.n_reset, zenc.n_reset])
    out['g3/vrnn/x_seed'] = x_seed
    np.savez_compressed(os.path.join(HERE, 'g2_g3_samplers.npz'), **out)
    print('G2/G3:', len(out), 'arrays')


def g4():
    from oracle import clvae_oracle as O
    out = {}
    ref = import_reference_pianoroll()
    P = ref.PianoData(os.path.join(REF, 'data/input/JSB Chorales_Cs.pickle'), batch_size=100, seq_length=1,
                      step_length=1, return_y_next=True, squeeze_x=True, squeeze_y=True)
    B = 16
    cfg = O.vae_config(latent_dim=4, n_classes=2, use_x_prev=True)
    rng = np.random.default_rng(42)
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in O.vae_init_params(cfg, seed=42).items()}
    x, xp = P.y_train[:B], P.x_train[:B]                       # cl_vae/train.py:58-60 wiring
    wt = np.eye(2)[P.train_song_keys[:B]]
    ew = rng.standard_normal((B, 1)).astype(np.float32).astype(np.float64)
    ez = rng.standard_normal((B, 4)).astype(np.float32).astype(np.float64)
    r = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)
    out.update({'vae/x': x.astype(np.uint8), 'vae/xp': xp.astype(np.uint8), 'vae/wt': wt, 'vae/ew': ew, 'vae/ez': ez})
    for k, v in p.items():
        out['vae/p/' + k] = v.astype(np.float32)
    for k, v in r['grads'].items():
        out['vae/g/' + k] = v
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo', 'acc'):
        out['vae/loss/' + k] = np.array(r[k])
    for k in ('h_w', 'w_mean', 'w_log_var', 'w', 'h', 'z_mean', 'z_log_var', 'z', 'h_dec', 'logits'):
        out['vae/c/' + k] = r['cache'][k]
    st = O.adam_wn_init(p)
    for _ in range(3):
        rr = O.vae_loss_and_grads(p, cfg, x, xp, wt, ew, ez)
        O.adam_wn_step(p, rr['grads'], st)
    for k, v in p.items():
        out['vae/p3/' + k] = v

    P2 = ref.PianoData(os.path.join(REF, 'data/input/JSB Chorales_all.pickle'), batch_size=200, seq_length=16,
                       step_length=1, return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)
    B, T = 8, 16
    cfg = O.vrnn_config(latent_dim=2, seq_length=T, n_classes=10, use_x_prev=True)
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in O.vrnn_init_params(cfg, seed=43).items()}
    idx = np.arange(0, 8 * 1300, 1300)
    X, Xp = P2.y_train[idx], P2.x_train[idx]                   # cl_vrnn/train.py:51-53 wiring
    wt = np.eye(10)[P2.train_song_keys[idx]]
    eW = rng.standard_normal((B, 9)).astype(np.float32).astype(np.float64)
    eZ = rng.standard_normal((B, T, 2)).astype(np.float32).astype(np.float64)
    r = O.vrnn_loss_and_grads(p, cfg, X, Xp, wt, eW, eZ)
    out.update({'vrnn/X': X.astype(np.uint8), 'vrnn/Xp': Xp.astype(np.uint8), 'vrnn/wt': wt, 'vrnn/eW': eW, 'vrnn/eZ': eZ})
    for k, v in p.items():
        out['vrnn/p/' + k] = v.astype(np.float32)
    for k, v in r['grads'].items():
        out['vrnn/g/' + k] = v.astype(np.float32) if v.size > 20000 else v
    for k in ('vae', 'kl_z', 'kl_w', 'w_rec', 'total', 'elbo', 'acc'):
        out['vrnn/loss/' + k] = np.array(r[k])
    for k in ('hW', 'Wargs', 'W', 'enc_h', 'Z_mean', 'Z_log_var', 'Z', 'dec_h', 'logits'):
        out['vrnn/c/' + k] = r['cache'][k]
    st = O.adam_wn_init(p)
    for _ in range(2):
        rr = O.vrnn_loss_and_grads(p, cfg, X, Xp, wt, eW, eZ)
        O.adam_wn_step(p, rr['grads'], st)
    for k, v in p.items():
        out['vrnn/p2/' + k] = v.astype(np.float32) if v.size > 20000 else v
    np.savez_compressed(os.path.join(HERE, 'g4_oracle_steps.npz'), **out)
    print('G4:', len(out), 'arrays')


if __name__ == '__main__':
    g1()
    g2_g3()
    g4()
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
