#!/usr/bin/env python
"""G8: the real `JSB Chorales_all` data set (the cl_vrnn scripts' real-data run: SURVEY.md 8d config 3, seq_length 16,
batch 200 -> 10400 / 3000 / 3000 windows, 10 classes) as a compact array fixture.
Runs ONLY in the build container (reads /root/reference/data/input); writes tests/golden/g8_jsb_all_notes.npz.

Data only, the layout of G7 (make_g7_jsb_cs.py): per split the MIDI note numbers of every frame (uint8, concatenated), the
number of notes per frame (uint8), the number of frames per song (uint16) and the songs' keys / modes.
tests/helpers.py::write_jsb_pickle('all', path) turns it back into a pickle of the reference's schema; the G1 checksums of
(JSB_all, batch 200, seq_length 16) in tests/golden/g1_pianodata.npz came from the reference's own loader on the original
pickle, and tests/test_pianodata.py holds the rebuilt pickle to them."""
import os
import pickle

import numpy as np

REF = '/root/reference/data/input/JSB Chorales_all.pickle'
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == '__main__':
    d = pickle.load(open(REF, 'rb'), encoding='latin1')
    out = {}
    for split in ('train', 'valid', 'test'):
        songs = d[split]
        out[split + '/notes'] = np.array([n for s in songs for f in s for n in f], dtype=np.uint8)
        out[split + '/per_frame'] = np.array([len(f) for s in songs for f in s], dtype=np.uint8)
        out[split + '/frames'] = np.array([len(s) for s in songs], dtype=np.uint16)
        out[split + '/key'] = np.array([str(k) for k in d[split + '_key']])
        out[split + '/mode'] = np.array([bool(m) for m in d[split + '_mode']])
        assert all(0 <= n < 256 for s in songs for f in s for n in f) and max(len(f) for s in songs for f in s) < 256
        assert max(len(s) for s in songs) < 65536
    path = os.path.join(HERE, 'g8_jsb_all_notes.npz')
    np.savez_compressed(path, **out)
    print('G8:', os.path.getsize(path) // 1024, 'KiB')
