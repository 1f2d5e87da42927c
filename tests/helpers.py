"""Shared test helpers (no GPU needed)."""
import os
import pickle

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_DATA = "/root/reference/data/input"


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def make_synthetic_pickle(path, n_songs=(12, 5, 5), seed=0, min_len=30, max_len=70, keys=('C', 'a', 'G', 'e', 'F')):
    """A pickle with the reference's schema: splits of songs = lists of
    timesteps = lists of MIDI ints 21..108, plus <split>_key (str) and <split>_mode (bool)."""
    rng = np.random.default_rng(seed)
    D = {}
    for split, n in zip(('train', 'valid', 'test'), n_songs):
        songs, ks, ms = [], [], []
        for i in range(n):
            L = int(rng.integers(min_len, max_len))
            base = int(rng.integers(40, 70))
            song = []
            for t in range(L):
                k = int(rng.integers(1, 5))
                song.append(sorted(set(int(base + d) for d in rng.integers(-12, 20, k))))
            songs.append(song)
            key = keys[i % len(keys)]
            ks.append(key)
            ms.append(key.isupper())
        D[split], D[split + '_key'], D[split + '_mode'] = songs, ks, ms
    with open(path, 'wb') as f:
        pickle.dump(D, f, protocol=2)
    return path


class StubModel:
    """Deterministic stand-in for a Keras sub-model; identical to the one in tests/golden/make_golden.py."""

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
        self.state = 0.5 * self.state + 0.01 * feat
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


def write_jsb_pickle(which, path):
    """A real JSB data set of the reference -- which = 'Cs' (`JSB Chorales_Cs`, tests/golden/g7_jsb_cs_notes.npz) or 'all'
    (`JSB Chorales_all`, g8_jsb_all_notes.npz), both made from the reference's pickles in the build container -- as a
    pickle of the reference's schema: {'train' | 'valid' | 'test': list[song], song: list[frame], frame: list[int MIDI];
    '<split>_key': list[str]; '<split>_mode': list[bool]}."""
    G = golden({'Cs': "g7_jsb_cs_notes.npz", 'all': "g8_jsb_all_notes.npz"}[which])
    D = {}
    for split in ('train', 'valid', 'test'):
        notes, per_frame, frames = G[split + '/notes'], G[split + '/per_frame'], G[split + '/frames']
        frame_end = np.cumsum(per_frame.astype(np.int64))
        frame_lists = [[int(n) for n in notes[e - c:e]] for e, c in zip(frame_end, per_frame)]
        song_end = np.cumsum(frames.astype(np.int64))
        D[split] = [frame_lists[e - c:e] for e, c in zip(song_end, frames)]
        D[split + '_key'] = [str(k) for k in G[split + '/key']]
        D[split + '_mode'] = [bool(m) for m in G[split + '/mode']]
    with open(path, 'wb') as f:
        pickle.dump(D, f, protocol=2)
    return path


def write_jsb_cs_pickle(path):
    return write_jsb_pickle('Cs', path)
