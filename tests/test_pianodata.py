"""G1: the product's PianoData vs tensors produced by the reference's own utils/pianoroll.py."""
import hashlib
import os

import numpy as np
import pytest

import clvae_amd  # noqa: F401
from clvae_amd.utils import pianoroll as PR
from helpers import REF_DATA, golden, make_synthetic_pickle, write_jsb_pickle

G1 = golden("g1_pianodata.npz")
CASES = [('Cs', 100, 1, dict(return_y_next=True, squeeze_x=True, squeeze_y=True)),
         ('Cs', 512, 1, dict(return_y_next=True, squeeze_x=True, squeeze_y=True)),
         ('all', 200, 16, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
         ('all', 256, 32, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
         ('all', 256, 64, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
         ('all', 256, 128, dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False)),
         ('Cs', 1, 32, dict(return_y_next=False, squeeze_x=False, squeeze_y=False)),
         ('all', None, 1, dict())]


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest()[:8], dtype=np.uint64)[0]


@pytest.mark.parametrize("name,bs,T,kw", CASES)
def test_pianodata_matches_reference_golden(name, bs, T, kw, tmp_path):
    """Both data sets are rebuilt from the committed note fixtures (G7: JSB_Cs, G8: JSB_all), so this runs anywhere and
    the fixtures themselves are held to the checksums the reference's loader produced on the reference's files."""
    path = write_jsb_pickle(name, str(tmp_path / ('JSB Chorales_%s.pickle' % name)))
    P = PR.PianoData(path, batch_size=bs, seq_length=T, step_length=1, **kw)
    tag = '%s_b%s_t%d' % (name, bs, T)
    for split in ('train', 'valid', 'test'):
        for xy in ('x', 'y'):
            a = getattr(P, '%s_%s' % (xy, split))
            assert a.dtype == np.float64
            assert tuple(G1['%s/%s_%s/shape' % (tag, xy, split)]) == a.shape
            assert float(G1['%s/%s_%s/sum' % (tag, xy, split)]) == a.sum()
            assert int(G1['%s/%s_%s/sha' % (tag, xy, split)]) == int(sha(a.astype(np.uint8)))
            if a.shape[0]:
                np.testing.assert_array_equal(G1['%s/%s_%s/head' % (tag, xy, split)], a[:4])
                np.testing.assert_array_equal(G1['%s/%s_%s/tail' % (tag, xy, split)], a[-4:])
        np.testing.assert_array_equal(G1['%s/%s_song_keys' % (tag, split)], getattr(P, '%s_song_keys' % split))
        np.testing.assert_array_equal(G1['%s/%s_song_inds' % (tag, split)], getattr(P, '%s_song_inds' % split))
        np.testing.assert_array_equal(G1['%s/%s_song_modes' % (tag, split)].astype(bool),
                                      getattr(P, '%s_song_modes' % split))
    keys = sorted(P.key_map, key=lambda k: P.key_map[k])
    assert [str(k) for k in G1['%s/key_map' % tag]] == keys


def test_helpers_match_reference_golden():
    np.testing.assert_array_equal(G1['helpers/song_to_pianoroll_low'], PR.song_to_pianoroll([(20, 30), (50,)]))
    np.testing.assert_array_equal(G1['helpers/song_to_pianoroll_high'], PR.song_to_pianoroll([(50, 109), (60,)]))
    np.testing.assert_array_equal(G1['helpers/sliding_window'],
                                  PR.sliding_window(np.arange(7 * 88).reshape(7, 88) % 5, 3, 2))
    assert PR.sliding_window(np.zeros((3, 88)), 3).size == 0           # n <= seq: no window (last one is dropped)
    assert PR.relative_major('a') == 'C' and PR.relative_major('E-') == 'E-'
    assert PR.pianoroll_to_song(PR.song_to_pianoroll([(60, 64), (62,)])) == [[60, 64], [62]]


def test_synthetic_schema_and_song_index_bug(tmp_path):
    f = make_synthetic_pickle(str(tmp_path / "syn.pickle"), min_len=10, max_len=40, seed=3)
    T = 20
    P = PR.PianoData(f, batch_size=4, seq_length=T, return_y_next=True, return_y_hist=True, squeeze_x=False,
                     squeeze_y=False)
    assert P.x_train.shape[1:] == (T, 88) and P.x_train.shape[0] % 4 == 0
    np.testing.assert_array_equal(P.x_train[:, 1:], P.y_train[:, :-1])     # y is x shifted by one frame
    # songs shorter than T+1 are dropped before indices are assigned (B2): the fix flag changes the lookups
    Pf = PR.PianoData(f, batch_size=4, seq_length=T, return_y_next=True, return_y_hist=True, squeeze_x=False,
                      squeeze_y=False, fix_song_index=True)
    np.testing.assert_array_equal(P.x_train, Pf.x_train)
    assert Pf.train_song_inds.max() >= P.train_song_inds.max()
    assert set(P.key_map) == {'C', 'G', 'F'}                               # a -> C, e -> G (relative major)
    P32 = PR.PianoData(f, batch_size=4, seq_length=T, dtype=np.float32)
    assert P32.x_train.dtype == np.float32


@pytest.mark.parametrize("kw", [dict(return_y_next=True, return_y_hist=True, squeeze_x=False, squeeze_y=False),
                                dict(return_y_next=True, return_y_hist=False, squeeze_x=False, squeeze_y=True),
                                dict(return_y_next=False, squeeze_x=False, squeeze_y=False)])
def test_lazy_windows_equal_the_materialised_arrays(tmp_path, kw):
    """PianoData(lazy=True): x_* / y_* are views of one uint8 frame store per split with exactly the
    values, shapes and song lookups of the eager arrays; row slices stay views."""
    f = make_synthetic_pickle(str(tmp_path / "syn.pickle"), min_len=10, max_len=60, seed=4)
    T = 12
    E = PR.PianoData(f, batch_size=8, seq_length=T, **kw)
    Lz = PR.PianoData(f, batch_size=8, seq_length=T, lazy=True, **kw)
    for split in ('train', 'valid', 'test'):
        for xy in ('x', 'y'):
            e, l = getattr(E, '%s_%s' % (xy, split)), getattr(Lz, '%s_%s' % (xy, split))
            assert tuple(e.shape) == tuple(l.shape) and len(e) == len(l)
            np.testing.assert_array_equal(e, np.asarray(l))
            assert np.asarray(l).dtype == np.float64
            if isinstance(l, PR.Windows) and len(l) > 8:
                sub = l[3:8]
                assert isinstance(sub, PR.Windows) and sub.store is l.store
                np.testing.assert_array_equal(e[3:8], np.asarray(sub))
                np.testing.assert_array_equal(e[5], l[5])
        np.testing.assert_array_equal(getattr(E, '%s_song_inds' % split), getattr(Lz, '%s_song_inds' % split))
        np.testing.assert_array_equal(getattr(E, '%s_song_keys' % split), getattr(Lz, '%s_song_keys' % split))
    assert isinstance(Lz.x_train, PR.Windows)
    assert Lz.x_train.store.dtype == np.uint8 and Lz.x_train.store.nbytes * T // 2 < E.x_train.nbytes // 8


@pytest.mark.parametrize("T,bs", [(2, 100), (3, 64)])
def test_flatten_windows_matches_the_reference_lines(T, bs, tmp_path):
    """G9: `cl_vae/train.py --seq_length T` (T > 1).  The reference's own lines cl_vae/train.py:21-30, run on windows from
    the reference's own loader on the real JSB_Cs, against this build's `flatten_windows` on the G7-rebuilt pickle: the
    note mask, `original_dim` and all six flattened arrays."""
    import types
    from clvae_amd.cl_vae.train import SPLITS, flatten_windows
    G9 = golden("g9_flatten_windows.npz")
    path = write_jsb_pickle('Cs', str(tmp_path / 'JSB Chorales_Cs.pickle'))
    P = PR.PianoData(path, batch_size=bs, seq_length=T, step_length=1, return_y_next=False, squeeze_x=True, squeeze_y=True)
    args = types.SimpleNamespace(seq_length=T, original_dim=88)
    flatten_windows(P, args)
    tag = 'Cs_b%d_t%d' % (bs, T)
    assert args.original_dim == int(G9[tag + '/original_dim']) == int(G9[tag + '/mask'].sum()) * T
    assert isinstance(args.original_dim, int)           # it goes into <run>.json
    for nm in SPLITS:
        a = getattr(P, nm)
        assert tuple(G9['%s/%s/shape' % (tag, nm)]) == a.shape and a.shape[1] == args.original_dim
        assert float(G9['%s/%s/sum' % (tag, nm)]) == a.sum()
        assert int(G9['%s/%s/sha' % (tag, nm)]) == int(sha(a.astype(np.uint8)))
        np.testing.assert_array_equal(G9['%s/%s/head' % (tag, nm)], a[:3])
        np.testing.assert_array_equal(G9['%s/%s/tail' % (tag, nm)], a[-3:])


def test_flatten_windows_refuses_single_frame_targets_like_the_reference(tmp_path):
    """`--seq_length 2 --use_x_prev` (or --predict_next): the targets are single frames [n, 88] next to windows [n, 2, 88]
    and the reference's np.vstack (cl_vae/train.py:22) raises ValueError (recorded in G9); so does this build."""
    import types
    from clvae_amd.cl_vae.train import flatten_windows
    G9 = golden("g9_flatten_windows.npz")
    assert str(G9['Cs_b100_t2_y_next/raises']) == 'ValueError'
    path = write_jsb_pickle('Cs', str(tmp_path / 'JSB Chorales_Cs.pickle'))
    P = PR.PianoData(path, batch_size=100, seq_length=2, step_length=1, return_y_next=True, squeeze_x=True, squeeze_y=True)
    assert P.x_train.shape == tuple(G9['Cs_b100_t2_y_next/x_shape']) and P.y_train.shape == tuple(G9['Cs_b100_t2_y_next/y_shape'])
    with pytest.raises(ValueError):
        flatten_windows(P, types.SimpleNamespace(seq_length=2, original_dim=88))
