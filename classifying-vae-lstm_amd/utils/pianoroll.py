"""Piano-roll data loader: pickle -> sliding-window tensors.

Mirrors the observable behaviour of the reference's utils/pianoroll.py (PianoData and its
helpers, :24-158), including its quirks, so the tensors are identical (golden G1):
  * a window of seq_length (+1 with return_y_next) frames starts at every index in
    range(n - window) -- the last possible window is dropped (:49-50);
  * songs that yield no window are removed BEFORE song indices are assigned, so per-window
    key/mode lookups index the unfiltered key list with a filtered position (:68-71, 147-152;
    SURVEY.md 5.9 B2) -- reproduced by default, `fix_song_index=True` repairs it;
  * rows are truncated to a multiple of batch_size (:154-158); split order train, test, valid.

Implementation is new: songs are rasterised once into a uint8 frame store and windows are
strided views gathered in one vectorised step (the reference stacks n float64 copies).
"""
import pickle

import numpy as np

rel_keys = {'a': 'C', 'b-': 'D-', 'b': 'D', 'c': 'E-', 'c#': 'E', 'd-': 'F-', 'd': 'F', 'd#': 'F#',
            'e-': 'G-', 'e': 'G', 'f': 'A-', 'f#': 'A', 'g': 'B-', 'g#': 'B', 'a-': 'C-'}


def relative_major(k):
    return k if k.isupper() else rel_keys[k]


def pianoroll_to_song(roll, offset=21):
    return [(np.where(x)[0] + offset).tolist() for x in roll]


def song_to_pianoroll(song, offset=21, dtype=np.float64):
    """[(60, 72, ...), ...] -> [n_frames, 88]; shifts the offset by an octave when notes fall outside."""
    all_notes = [y for x in song for y in x]
    if min(all_notes) - offset < 0:
        offset -= 12
    if max(all_notes) - offset > 87:
        offset += 12
    roll = np.zeros((len(song), 88), dtype=dtype)
    for i, notes in enumerate(song):
        roll[i, [n - offset for n in notes]] = 1
    return roll


def sliding_inds(n, seq_length, step_length):
    return np.arange(n - seq_length, step=step_length)


def sliding_window(roll, seq_length, step_length=1):
    """[n_win, seq_length, 88]; with step 1, out[i,1:] == out[i+1,:-1]."""
    starts = sliding_inds(roll.shape[0], seq_length, step_length)
    if len(starts) == 0:
        return np.array([])
    return roll[starts[:, None] + np.arange(seq_length)[None, :]]


def songs_to_pianoroll(songs, seq_length, step_length, inner_fcn=song_to_pianoroll):
    rolls = [sliding_window(inner_fcn(s), seq_length, step_length) for s in songs]
    rolls = [r for r in rolls if len(r) > 0]
    inds = [i * np.ones((len(r),)) for i, r in enumerate(rolls)]
    return np.vstack(rolls), np.hstack(inds)


class Windows:
    """Read-only [n, length, 88] view of a uint8 frame store: row i is store[starts[i] + t0 : starts[i] + t0 + length].
    The sliding windows of a split overlap almost entirely (step 1: 128 copies of every frame at seq_length 128), so
    PianoData(lazy=True) hands these out instead of arrays: numpy sees an array (np.asarray materialises it), row
    slicing stays lazy, and Model.fit uploads the store once and gathers windows by start offset on the device
    (SURVEY.md 8f4)."""

    def __init__(self, store, starts, t0, length, dtype=np.float64):
        self.store, self.starts = store, np.asarray(starts, dtype=np.int64)
        self.t0, self.length, self.dtype = int(t0), int(length), np.dtype(dtype)

    @property
    def shape(self):
        return (len(self.starts), self.length, self.store.shape[1])

    ndim = 3

    def __len__(self):
        return len(self.starts)

    def __array__(self, dtype=None, copy=None):
        idx = (self.starts + self.t0)[:, None] + np.arange(self.length)[None, :]
        return self.store[idx].astype(dtype or self.dtype)

    def __getitem__(self, key):
        if isinstance(key, (slice, np.ndarray, list)):          # rows: still a view
            return Windows(self.store, self.starts[key], self.t0, self.length, self.dtype)
        return np.asarray(self)[key]

    def squeeze(self):
        return np.asarray(self).squeeze() if 1 in self.shape else self

    def same_frames(self, other):
        """True when `other` shows exactly the same frames (same store, offsets and length)."""
        return (self.store is other.store and self.t0 == other.t0 and self.length == other.length
                and np.array_equal(self.starts, other.starts))


def _load_pickle(path):
    with open(path, 'rb') as f:
        try:
            return pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            return pickle.load(f, encoding='latin1')      # Python-2 pickles of the reference


class PianoData:
    def __init__(self, train_file, batch_size=None, seq_length=1, step_length=1, return_y_next=True,
                 return_y_hist=False, squeeze_x=True, squeeze_y=True, use_rel_major=True,
                 fix_song_index=False, dtype=np.float64, lazy=False):
        """lazy=True: x_* / y_* are `Windows` views of one uint8 frame store per split instead of materialised arrays
        (same values: np.asarray(view) == the eager array)."""
        D = _load_pickle(train_file)
        self.lazy = lazy
        self.train_file = train_file
        self.batch_size = batch_size
        self.seq_length = seq_length
        self.step_length = step_length
        self.return_y_next = return_y_next
        self.return_y_hist = return_y_hist
        self.squeeze_x = squeeze_x
        self.squeeze_y = squeeze_y
        self.use_rel_major = use_rel_major
        self.fix_song_index = fix_song_index
        self.dtype = dtype

        self.x_train, self.y_train, self.train_song_inds = self.make_xy(D['train'])
        self.x_test, self.y_test, self.test_song_inds = self.make_xy(D['test'])
        self.x_valid, self.y_valid, self.valid_song_inds = self.make_xy(D['valid'])

        if 'train_mode' in D:
            self.train_song_modes = self.song_modes(D['train_mode'], self.train_song_inds)
            self.test_song_modes = self.song_modes(D['test_mode'], self.test_song_inds)
            self.valid_song_modes = self.song_modes(D['valid_mode'], self.valid_song_inds)
        if 'train_key' in D:
            D = self.update_keys(D)
            self.key_map = self.make_keymap(D)
            self.train_song_keys = self.song_keys(D['train_key'], self.train_song_inds)
            self.test_song_keys = self.song_keys(D['test_key'], self.test_song_inds)
            self.valid_song_keys = self.song_keys(D['valid_key'], self.valid_song_inds)

    def make_xy(self, songs):
        win = self.seq_length + int(self.return_y_next)
        if self.lazy:
            return self._make_xy_lazy(songs, win)
        rolls, inds = [], []
        kept = 0
        for si, s in enumerate(songs):
            r = sliding_window(song_to_pianoroll(s, dtype=np.uint8), win, self.step_length)
            if len(r) == 0:
                continue
            rolls.append(r)
            # the reference numbers songs AFTER dropping the empty ones (B2)
            inds.append(np.full(len(r), si if self.fix_song_index else kept, dtype=np.float64))
            kept += 1
        x_rolls = np.vstack(rolls).astype(self.dtype)
        song_inds = np.hstack(inds)
        x_rolls = self.adjust_for_batch_size(x_rolls)
        song_inds = self.adjust_for_batch_size(song_inds)
        if self.return_y_next:
            y_rolls = x_rolls[:, 1:, :] if self.return_y_hist else x_rolls[:, -1, :]
            x_rolls = x_rolls[:, :-1, :]
        else:
            y_rolls = x_rolls
        if self.squeeze_x:
            x_rolls = x_rolls.squeeze()
        if self.squeeze_y:
            y_rolls = y_rolls.squeeze()
        return x_rolls, y_rolls, song_inds

    def _make_xy_lazy(self, songs, win):
        """The same windows as make_xy, as views: every song is rasterised once into a shared uint8 store and a window
        is its start frame."""
        store, starts, inds = [], [], []
        kept = frames = 0
        for si, s in enumerate(songs):
            roll = song_to_pianoroll(s, dtype=np.uint8)
            st = sliding_inds(roll.shape[0], win, self.step_length)
            if len(st) > 0:
                starts.append(frames + st)
                inds.append(np.full(len(st), si if self.fix_song_index else kept, dtype=np.float64))
                kept += 1
            store.append(roll)
            frames += roll.shape[0]
        store = np.vstack(store)
        starts = self.adjust_for_batch_size(np.hstack(starts))
        song_inds = self.adjust_for_batch_size(np.hstack(inds))
        T = self.seq_length
        if self.return_y_next:
            x = Windows(store, starts, 0, T, self.dtype)
            y = Windows(store, starts, 1, T, self.dtype) if self.return_y_hist else Windows(store, starts, T, 1, self.dtype)
        else:
            x = y = Windows(store, starts, 0, win, self.dtype)
        if self.squeeze_x:
            x = x.squeeze()
        if self.squeeze_y:
            y = y.squeeze()
        return x, y, song_inds

    def song_modes(self, modes, song_inds):
        return np.array(modes)[song_inds.astype(int)]

    def update_keys(self, D):
        if not self.use_rel_major:
            return D
        for k in ('train_key', 'test_key', 'valid_key'):
            D[k] = [relative_major(x) for x in D[k]]
        return D

    def make_keymap(self, D):
        all_keys = np.unique(np.hstack([D['train_key'], D['test_key'], D['valid_key']]))
        return dict(zip([str(k) for k in all_keys], range(len(all_keys))))

    def song_keys(self, keys, song_inds):
        key_inds = [self.key_map[k] for k in keys]
        return np.array(key_inds)[song_inds.astype(int)]

    def adjust_for_batch_size(self, items):
        if self.batch_size is None:
            return items
        mod = items.shape[0] % self.batch_size
        return items[:-mod] if mod > 0 else items
