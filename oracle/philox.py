"""numpy restatement of the Philox4x32-10 draw used by clv_philox_{normal,uniform} (TEST ORACLE).

Test infrastructure only (see oracle/__init__.py).  The reference draws its noise from TF's
RNG (K.random_normal, cl_vae/model.py:152,172; cl_vrnn/model.py:185,214), which cannot be
reproduced; the build replaces it by a counter-based generator so that a sample's noise is a
pure function of (seed, step, stream_id, global element index).  Philox4x32-10 as published by
Salmon et al. (SC'11): multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 / 0xBB67AE85.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + W0); k1 = np.uint32(k1 + W1)
    return c0, c1, c2, c3


def _u01(x):
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def _words(n, seed, step, stream_id, first_index):
    idx = np.uint64(first_index) + np.arange(n, dtype=np.uint64)
    ctr = idx >> np.uint64(2)
    r = philox4x32_10((ctr & MASK).astype(np.uint32), (ctr >> np.uint64(32)).astype(np.uint32),
                      np.full(n, stream_id, np.uint32), np.full(n, step, np.uint32),
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    return idx, np.stack(r, axis=1)


def uniform(n, seed, step=0, stream_id=0, first_index=0):
    idx, r = _words(n, seed, step, stream_id, first_index)
    return _u01(r[np.arange(n), (idx & np.uint64(3)).astype(np.int64)])


def normal(n, seed, step=0, stream_id=0, first_index=0):
    idx, r = _words(n, seed, step, stream_id, first_index)
    j = (idx & np.uint64(3)).astype(np.int64)
    pair = j >> 1
    u1 = _u01(r[np.arange(n), 2 * pair]).astype(np.float64)
    u2 = _u01(r[np.arange(n), 2 * pair + 1]).astype(np.float64)
    rad = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return np.where((j & 1) == 0, rad * np.cos(th), rad * np.sin(th)).astype(np.float32)
