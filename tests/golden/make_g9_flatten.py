#!/usr/bin/env python
"""Regenerates tests/golden/g9_flatten_windows.npz.  Runs ONLY in the build container (needs /root/reference).

G9  what `cl_vae/train.py --seq_length T` (T > 1) feeds the model: the reference's own `utils/pianoroll.py` builds the
    windows from the real `JSB Chorales_Cs.pickle` (Python-2 shims as in make_golden.py) and the reference's own lines
    cl_vae/train.py:21-30 (exec'd as they stand, dedented) flatten them to the sounding notes side by side and set
    `args.original_dim`.  Stored: the note mask, original_dim, and shape / sum / checksum / first and last rows of the
    six flattened arrays -- arrays only, no reference text.
"""
import os
import sys
import textwrap
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, checksum, import_reference_pianoroll  # noqa: E402


def main():
    ref = import_reference_pianoroll()
    with open(os.path.join(REF, 'code', 'cl_vae', 'train.py')) as f:
        body = textwrap.dedent(''.join(f.readlines()[20:30]))        # lines 21-30
    out = {}
    for T, bs, kw in ((2, 100, dict(return_y_next=False)), (3, 64, dict(return_y_next=False))):
        P = ref.PianoData(os.path.join(REF, 'data', 'input', 'JSB Chorales_Cs.pickle'), batch_size=bs, seq_length=T,
                          step_length=1, squeeze_x=True, squeeze_y=True, **kw)
        args = types.SimpleNamespace(seq_length=T, original_dim=88)
        X = np.vstack([P.x_train, P.x_valid, P.x_test, P.y_train, P.y_valid, P.y_test])
        mask = X.sum(axis=0).sum(axis=0) > 0
        exec(compile(body, 'cl_vae/train.py:21-30', 'exec'), {'np': np, 'P': P, 'args': args})
        tag = 'Cs_b%d_t%d' % (bs, T)
        out[tag + '/mask'] = mask
        out[tag + '/original_dim'] = np.array(int(args.original_dim))
        for nm in ('x_train', 'x_valid', 'x_test', 'y_train', 'y_valid', 'y_test'):
            a = getattr(P, nm)
            out['%s/%s/shape' % (tag, nm)] = np.array(a.shape)
            out['%s/%s/sum' % (tag, nm)] = np.array(a.sum())
            out['%s/%s/sha' % (tag, nm)] = np.array(checksum(a.astype(np.uint8)))
            out['%s/%s/head' % (tag, nm)] = a[:3].astype(np.uint8)
            out['%s/%s/tail' % (tag, nm)] = a[-3:].astype(np.uint8)
        print(tag, 'original_dim', int(args.original_dim), 'x_train', P.x_train.shape)
    # with --use_x_prev / --predict_next the targets are single frames [n, 88] next to windows [n, T, 88]: the reference's
    # own np.vstack (cl_vae/train.py:22) refuses them -- recorded so that the build keeps the same refusal
    P = ref.PianoData(os.path.join(REF, 'data', 'input', 'JSB Chorales_Cs.pickle'), batch_size=100, seq_length=2,
                      step_length=1, squeeze_x=True, squeeze_y=True, return_y_next=True)
    try:
        exec(compile(body, 'cl_vae/train.py:21-30', 'exec'), {'np': np, 'P': P, 'args': types.SimpleNamespace(seq_length=2)})
        raised = ''
    except Exception as e:
        raised = type(e).__name__
    print('return_y_next=True:', raised, P.x_train.shape, P.y_train.shape)
    out['Cs_b100_t2_y_next/raises'] = np.array(raised)
    out['Cs_b100_t2_y_next/x_shape'] = np.array(P.x_train.shape)
    out['Cs_b100_t2_y_next/y_shape'] = np.array(P.y_train.shape)
    np.savez_compressed(os.path.join(HERE, 'g9_flatten_windows.npz'), **out)


if __name__ == '__main__':
    main()
