#!/usr/bin/env python
"""G7: the real `JSB Chorales_Cs` data set (BASELINE config 1's --train_file) as a compact array fixture.
Runs ONLY in the build container (reads /root/reference/data/input); writes tests/golden/g7_jsb_cs_notes.npz.

Data only: per split the MIDI note numbers of every frame (uint8, concatenated), the number of notes per frame (uint8),
the number of frames per song (uint16) and the songs' keys / modes -- everything the reference's pickle holds, in ~1/30 of
its 2 MB.  tests/helpers.py::write_jsb_cs_pickle() turns it back into a pickle of the reference's schema, so that the train
CLI runs on the real 13,807 / 4,602 / 4,725 frames on the GPU box, where /root/reference does not exist."""
import os
import pickle

import numpy as np

REF = '/root/reference/data/input/JSB Chorales_Cs.pickle'
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
    path = os.path.join(HERE, 'g7_jsb_cs_notes.npz')
    np.savez_compressed(path, **out)
    print('G7:', os.path.getsize(path) // 1024, 'KiB')
