"""Build-container / any-CPU tool (no GPU): how far do two correct trajectories of the JSB_all run drift apart?  fp64 oracle vs the same oracle in float32 (what Keras' floatx computes)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import write_jsb_pickle
from oracle import clvae_oracle as O
from oracle import philox as OP
import clvae_amd
from clvae_amd.utils.pianoroll import PianoData
path = write_jsb_pickle('all', '/tmp/jsb_all.pickle')
B, T, L, Cn, E, seed = 200, 16, 2, 10, 3, 2025
P = PianoData(path, batch_size=B, seq_length=T, step_length=1, return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)
cfg = O.vrnn_config(latent_dim=L, seq_length=T, n_classes=Cn, use_x_prev=True)
p0 = {k: np.asarray(v, np.float32) for k, v in O.vrnn_init_params(cfg, seed=5).items()}
cur, hst, wt = P.y_train, P.x_train, np.eye(Cn)[P.train_song_keys.astype(int)]
vcur, vhst, vwt = P.y_valid, P.x_valid, np.eye(Cn)[P.valid_song_keys.astype(int)]
keys = ('total', 'vae', 'kl_w', 'w_rec', 'kl_z', 'acc')
def run(dt):
    p = {k: v.astype(dt) for k, v in p0.items()}
    st = O.adam_wn_init(p)
    np.random.seed(12); it = 0; out = []
    for ep in range(E):
        index = np.arange(len(cur)); np.random.shuffle(index)
        acc = np.zeros(6)
        for b0 in range(0, len(cur), B):
            rows = index[b0:b0 + B]
            ew = OP.normal(B * (Cn - 1), seed, step=it, stream_id=0).reshape(B, Cn - 1).astype(np.float32).astype(dt)
            ez = OP.normal(B * T * L, seed, step=it, stream_id=1).reshape(B, T, L).astype(np.float32).astype(dt)
            r = O.vrnn_loss_and_grads(p, cfg, cur[rows].astype(dt), hst[rows].astype(dt), wt[rows].astype(dt), ew, ez)
            O.adam_wn_step(p, r['grads'], st)
            acc += [r[k] for k in keys]; it += 1
        tr = acc / (len(cur) // B)
        acc = np.zeros(6)
        for j, b0 in enumerate(range(0, len(vcur), B)):
            ew = OP.normal(B * (Cn - 1), seed, step=it, stream_id=2 * (2 + j)).reshape(B, Cn - 1).astype(np.float32).astype(dt)
            ez = OP.normal(B * T * L, seed, step=it, stream_id=2 * (2 + j) + 1).reshape(B, T, L).astype(np.float32).astype(dt)
            r = O.vrnn_loss_and_grads(p, cfg, vcur[b0:b0 + B].astype(dt), vhst[b0:b0 + B].astype(dt), vwt[b0:b0 + B].astype(dt), ew, ez, need_grads=False)
            acc += [r[k] for k in keys]
        out.append(np.concatenate([tr, acc / (len(vcur) // B)]))
    return np.array(out, np.float64), {k: v.astype(np.float64) for k, v in p.items()}
t = time.time()
(a, pa), (b, pb) = run(np.float64), run(np.float32)
print("seconds", time.time() - t)
np.set_printoptions(linewidth=200, precision=3)
print("columns: train (total vae kl_w w_rec kl_z acc) | val (same)")
print("fp64:\n", a)
print("|fp32 - fp64|:\n", np.abs(a - b))
print("parameters after %d steps, float32 oracle against float64 oracle: fraction of entries beyond rtol 2e-3 / atol 5e-5, max |dw|" % (E * (len(cur) // B)))
for k in pa:
    d = np.abs(pa[k] - pb[k])
    print("  %-28s %.3e  %.3e" % (k, float((d > 2e-3 * np.abs(pa[k]) + 5e-5).mean()), float(d.max())))
