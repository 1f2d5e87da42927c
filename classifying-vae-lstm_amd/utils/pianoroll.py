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


SPLITS = ('train', 'test', 'valid')          # the order the reference builds them in


class _WindowPlan:
    """Where the windows of one split start.  Every song is rasterised ONCE into a shared uint8 frame store; a window is
    a start frame in it.  Songs too short for a window contribute frames but no starts, and -- the reference's quirk B2
    (utils/pianoroll.py:68-71: indices are assigned after the empty songs are dropped) -- the song number of a window
    counts only the songs that did yield windows unless fix_song_index is set."""

    def __init__(self, songs, window, step, fix_song_index):
        rolls, starts, owner = [], [], []
        first_frame = counted = 0
        for number, song in enumerate(songs):
            roll = song_to_pianoroll(song, dtype=np.uint8)
            begin = sliding_inds(roll.shape[0], window, step)
            if len(begin):
                starts.append(first_frame + begin)
                owner.append(np.full(len(begin), number if fix_song_index else counted, dtype=np.float64))
                counted += 1
            rolls.append(roll)
            first_frame += roll.shape[0]
        self.store = np.vstack(rolls)
        self.starts = np.hstack(starts)
        self.song_of_window = np.hstack(owner)
        self.window = window

    def keep_whole_batches(self, batch_size):
        """rows beyond the last full batch are dropped (utils/pianoroll.py:154-158)"""
        if batch_size is not None:
            n = len(self.starts) - len(self.starts) % batch_size
            self.starts, self.song_of_window = self.starts[:n], self.song_of_window[:n]


class PianoData:
    """The reference's loader (utils/pianoroll.py:73-158) by attribute: x_<split>, y_<split>, <split>_song_inds and, when the
    pickle has them, <split>_song_modes, <split>_song_keys and key_map, for split in train / test / valid.

      x: the first seq_length frames of a window; y: the frame after them (return_y_next), or the window shifted by one
      frame (return_y_hist), or x itself (return_y_next=False).
      lazy=True: x_* / y_* are `Windows` views of one uint8 frame store per split instead of materialised arrays (same
      values: np.asarray(view) == the eager array)."""

    def __init__(self, train_file, batch_size=None, seq_length=1, step_length=1, return_y_next=True,
                 return_y_hist=False, squeeze_x=True, squeeze_y=True, use_rel_major=True,
                 fix_song_index=False, dtype=np.float64, lazy=False):
        for name, value in list(locals().items()):          # every option is a public attribute, like in the reference
            if name != 'self':
                setattr(self, name, value)
        pickled = _load_pickle(train_file)
        for split in SPLITS:
            x, y, owner = self._windows_of(pickled[split])
            setattr(self, 'x_' + split, x)
            setattr(self, 'y_' + split, y)
            setattr(self, split + '_song_inds', owner)
        self._label_windows(pickled)

    # -- frames ---------------------------------------------------------------
    def _windows_of(self, songs):
        T, nxt = self.seq_length, int(self.return_y_next)
        plan = _WindowPlan(songs, T + nxt, self.step_length, self.fix_song_index)
        plan.keep_whole_batches(self.batch_size)
        # (first frame, length) of x and y inside a window
        x_span = (0, T) if nxt else (0, plan.window)
        y_span = x_span if not nxt else ((1, T) if self.return_y_hist else (T, 1))
        view = lambda span: Windows(plan.store, plan.starts, span[0], span[1], self.dtype)
        x, y = view(x_span), (view(y_span) if y_span != x_span else None)
        if not self.lazy:
            x = np.asarray(x)
            y = x if y is None else np.asarray(y)
        elif y is None:
            y = x
        if nxt and not self.return_y_hist and not self.lazy:
            y = y[:, 0, :]                                   # the reference slices one frame out: [n, 88], not [n, 1, 88]
        if self.squeeze_x:
            x = x.squeeze()
        if self.squeeze_y:
            y = y.squeeze()
        return x, y, plan.song_of_window

    # -- labels ---------------------------------------------------------------
    def _label_windows(self, pickled):
        """Per-window mode and key of the song a window came from; keys as integers through key_map (the sorted union of
        the three splits' keys, minor keys folded onto their relative major when use_rel_major)."""
        per_window = lambda values, split: np.array(values)[getattr(self, split + '_song_inds').astype(int)]
        if 'train_mode' in pickled:
            for split in SPLITS:
                setattr(self, split + '_song_modes', per_window(pickled[split + '_mode'], split))
        if 'train_key' in pickled:
            fold = relative_major if self.use_rel_major else (lambda k: k)
            keys = {split: [fold(k) for k in pickled[split + '_key']] for split in SPLITS}
            names = np.unique(np.hstack([keys[split] for split in SPLITS]))
            self.key_map = {str(k): i for i, k in enumerate(names)}
            for split in SPLITS:
                setattr(self, split + '_song_keys', per_window([self.key_map[k] for k in keys[split]], split))
