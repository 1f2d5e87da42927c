"""-m gpu: BASELINE configuration 1 end to end on the REAL data.

`cl_vae/train.py run1 --use_x_prev --latent_dim 4 --train_file 'JSB Chorales_Cs.pickle'` (cl_vae/train.py:13-74; batch 100,
13,300 training and 4,400 validation frames: 133 + 44 batches per epoch) runs on the device for three epochs through the
train CLI's own `train(args)`, on the data set rebuilt from the committed fixture G7.  An oracle-driven loop replays the
run on the CPU: the same initial weights (initializers.init_weights at the run's seed), the same epoch permutations
(np.random.shuffle under the same global seed, Keras' A.4), the same Philox noise (oracle/philox.py at (seed, step =
iterations, streams 0/1; validation chunk j: streams 4+2j / 5+2j, SURVEY.md B10: the sampling noise stays on)),
`vae_loss_and_grads` + `adam_wn_step` per batch.  Compared: every History key of every epoch, which epoch the checkpoint
callback kept (utils/model_utils.py:106-158: best val_loss from epoch 1 on), the weights in the .h5 it wrote, the final
weights, the entries train() returns."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import write_jsb_cs_pickle
from oracle import clvae_oracle as O
from oracle import philox as OP

pytestmark = pytest.mark.gpu


def f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def test_config1_train_cli_on_jsb_cs_tracks_an_oracle_loop(tmp_path):
    import clvae_amd  # noqa: F401
    from clvae_amd import _lib
    from clvae_amd.cl_vae import train as TR
    from clvae_amd.initializers import init_weights
    from clvae_amd.utils import h5io
    from clvae_amd.utils.pianoroll import PianoData
    _lib.require_gpu()
    path = write_jsb_cs_pickle(str(tmp_path / 'JSB Chorales_Cs.pickle'))
    E, B, L, seed, np_seed = 3, 100, 4, 2024, 11
    args = TR.build_parser().parse_args(['run1', '--use_x_prev', '--latent_dim', str(L), '--num_epochs', str(E),
                                         '--train_file', path, '--model_dir', str(tmp_path)])
    assert args.batch_size == B and args.optimizer == 'adam-wn' and args.patience == 5      # the reference's defaults
    args.seed = seed
    np.random.seed(np_seed)
    model, best = TR.train(args)
    torch.cuda.synchronize()
    hist = model.history.history
    assert model.engine.fused and int(model.engine.P.iterations.item()) == E * 133

    # ---- the same run on the oracle ------------------------------------------------------------------------------------
    P = PianoData(path, batch_size=B, seq_length=1, step_length=1, return_y_next=True, squeeze_x=True, squeeze_y=True)
    assert P.x_train.shape == (13300, 88) and P.x_valid.shape == (4400, 88) and args.n_classes == 2
    cfg = O.vae_config(latent_dim=L, n_classes=2, use_x_prev=True)
    p = {k: f32(v) for k, v in init_weights(model.engine.P.logical, model.engine.cfg, seed=seed).items()}
    cur, hst, wt = P.y_train, P.x_train, np.eye(2)[P.train_song_keys]          # cl_vae/train.py:58-66: inputs [y, x]
    vcur, vhst, vwt = P.y_valid, P.x_valid, np.eye(2)[P.valid_song_keys]
    st = O.adam_wn_init(p)
    names = ('x_decoded_mean', 'w', 'w2', 'z_args')                            # outputs: recon, kl_w, w_rec, kl_z
    keys = ('vae', 'kl_w', 'w_rec', 'kl_z')
    ref, snaps, it = {}, [], 0
    np.random.seed(np_seed)
    for ep in range(E):
        index = np.arange(len(cur))
        np.random.shuffle(index)
        acc = np.zeros(6)
        for b0 in range(0, len(cur), B):
            rows = index[b0:b0 + B]
            ew = f32(OP.normal(B, seed, step=it, stream_id=0).reshape(B, 1))
            ez = f32(OP.normal(B * L, seed, step=it, stream_id=1).reshape(B, L))
            r = O.vae_loss_and_grads(p, cfg, cur[rows], hst[rows], wt[rows], ew, ez)
            O.adam_wn_step(p, r['grads'], st)
            acc += [r['total']] + [r[k] for k in keys] + [r['acc']]
            it += 1
        logs = dict(zip(['loss'] + [n + '_loss' for n in names] + ['w_acc'], acc / (len(cur) // B)))
        acc = np.zeros(6)
        for j, b0 in enumerate(range(0, len(vcur), B)):
            ew = f32(OP.normal(B, seed, step=it, stream_id=2 * (2 + j)).reshape(B, 1))
            ez = f32(OP.normal(B * L, seed, step=it, stream_id=2 * (2 + j) + 1).reshape(B, L))
            r = O.vae_loss_and_grads(p, cfg, vcur[b0:b0 + B], vhst[b0:b0 + B], vwt[b0:b0 + B], ew, ez, need_grads=False)
            acc += [r['total']] + [r[k] for k in keys] + [r['acc']]
        logs.update(dict(zip(['val_loss'] + ['val_' + n + '_loss' for n in names] + ['val_w_acc'], acc / (len(vcur) // B))))
        for k, v in logs.items():
            ref.setdefault(k, []).append(v)
        snaps.append({k: v.copy() for k, v in p.items()})

    # ---- History: every key, every epoch -------------------------------------------------------------------------------
    assert set(hist) == set(ref)
    worst = 0.0
    for k in sorted(ref):
        got, want = np.asarray(hist[k], np.float64), np.asarray(ref[k])
        assert got.shape == (E,)
        tol = 2e-3 if k.endswith('acc') else 1e-3          # losses: nats per frame, the north star's ELBO tolerance
        assert np.abs(got - want).max() <= tol, (k, got, want)
        worst = max(worst, float(np.abs(got - want).max()))
    print("config 1, %d epochs on JSB_Cs: History max |gpu - oracle| %.2e; val_loss %s" % (E, worst, np.round(hist['val_loss'], 4)))

    # ---- the checkpoint: best val_loss from epoch 1 on (epoch 0 is never kept: min_epoch = max(anneals) + 1) ---------
    at = 1 + int(np.argmin(ref['val_loss'][1:]))
    assert at == 1 + int(np.argmin(hist['val_loss'][1:]))
    for k, v in best.items():
        assert v == hist[k][at]
    saved = dict()
    for lname, ws in h5io.load_keras_weights(os.path.join(str(tmp_path), 'run1.h5')):
        layer = model.get_layer(lname)
        for wn, arr in zip(layer.weight_names, ws):
            saved['%s/%s' % (lname, wn)] = np.asarray(arr, np.float64)
    final = model.engine.P.get_weights()
    assert set(saved) == set(final) == set(p)

    def far(a, b, steps):      # see tests/test_gpu_timed_step.py: Adam moves an entry by +-lr whatever its gradient's size
        d = np.abs(a - b)
        return float((d > 2e-3 * np.abs(b) + 5e-5).mean()), float(d.max()), 2e-3 * steps
    wf = wm = 0.0
    for k in p:
        for a, b, steps in ((saved[k], snaps[at][k], (at + 1) * 133), (final[k], p[k], E * 133)):
            frac, dmax, cap = far(a, b, steps)
            assert frac <= 2e-2 and dmax <= cap, (k, frac, dmax)
            wf, wm = max(wf, frac), max(wm, dmax)
    print("config 1: checkpoint = epoch %d; parameters after up to %d steps: worst tensor has %.2e of its entries beyond "
          "rtol 2e-3 / atol 5e-5, max |dw| %.2e" % (at, E * 133, wf, wm))
    assert os.path.exists(os.path.join(str(tmp_path), 'run1.json')) and os.path.exists(os.path.join(str(tmp_path), 'run1.yaml'))
