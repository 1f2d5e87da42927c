"""Piano-roll -> Standard MIDI File writer, dependency-free.

Mirrors utils/midi_utils.py of the reference (MidiWriter.dump_sequence_to_midi :26-98,
write_sample :100-104), which needs the third-party `midi` (python-midi) package -- absent here,
so the byte encoding below restates python-midi's FileWriter from memory:
format-1 header, `resolution` ticks per quarter, a meta track holding one TimeSignature event
(4/4, metronome 24, 8 thirty-seconds), then the note track; variable-length delta times; running
status (a status byte is emitted only when it changes); python-midi appends no End-of-Track
event, `end_of_track=True` adds the two that the SMF standard asks for.

Event semantics follow the reference loop exactly: a frame lasts `time_step` ticks; per frame,
note-offs (ascending pitch) come before note-ons (ascending pitch); only the first event of a
frame carries the accumulated delta; a note sounding in consecutive frames is held; notes still
on at the end are released; pitch = column + offset (21); velocity 100.
"""
import os
import struct

import numpy as np

RANGE = 128


def write_varlen(value):
    out = [value & 0x7F]
    value >>= 7
    while value:
        out.append((value & 0x7F) | 0x80)
        value >>= 7
    return bytes(reversed(out))


class MidiWriter(object):
    """Piano-roll -> SMF bytes.  Events are (delta ticks, status byte, data bytes) triples."""

    def __init__(self, verbose=False, default_vel=100):
        self.verbose = verbose
        self.note_range = RANGE
        self.default_velocity = default_vel

    @staticmethod
    def _encode_track(events, end_of_track):
        buf = bytearray()
        running = None
        for tick, status, data in events:
            buf += write_varlen(int(tick))
            if status == 0xFF:                       # meta: FF type len data, never running status
                buf += bytes([0xFF, data[0]]) + write_varlen(len(data) - 1) + bytes(data[1:])
            else:
                if running != status:
                    running = status
                    buf.append(status)
                buf += bytes(data)
        if end_of_track:
            buf += b'\x00\xFF\x2F\x00'
        return b'MTrk' + struct.pack('>I', len(buf)) + bytes(buf)

    def note_events(self, roll, time_step, offset):
        """The note track of a [frames, notes] roll as the difference of consecutive frames: what stopped sounding is
        released (ascending pitch), then what started is struck (ascending pitch); after the last frame everything
        still sounding is released.  The ticks since the previous event ride on a frame's FIRST event, the others of
        the frame follow at delta 0; a frame without changes only lets the ticks accumulate."""
        sounding = np.asarray(roll) != 0
        silent = np.zeros((1, sounding.shape[1]), dtype=bool)
        before = np.vstack([silent, sounding])           # frame f-1 (nothing before the first; all released at the end)
        after = np.vstack([sounding, silent])
        events, pending = [], 0
        for prev, cur in zip(before, after):
            pending += time_step
            released = np.flatnonzero(prev & ~cur) + offset
            struck = np.flatnonzero(cur & ~prev) + offset
            for status, pitches, vel in ((0x80, released, 0), (0x90, struck, self.default_velocity)):
                for pitch in pitches.tolist():
                    events.append((pending, status, (pitch, vel)))
                    pending = 0
        return events

    def dump_sequence_to_midi(self, seq, output_filename, time_step=120, resolution=480, metronome=24, offset=21,
                              format='final', end_of_track=False):
        """format 'final': [frames, notes] 0/1 roll; 'icml': a list of note lists per frame; 'flat': one long vector."""
        if format == 'icml':
            roll = np.zeros((len(seq), self.note_range))
            for f, notes in enumerate(seq):
                roll[f, [n for n in notes if 0 <= n < self.note_range]] = 1
        elif format == 'flat':
            roll = np.reshape(seq, [-1, self.note_range])
        else:
            roll = np.asarray(seq)
        self.track = self.note_events(roll, time_step, offset)
        meta_track = [(0, 0xFF, (0x58, 4, 2, metronome, 8))]          # 4/4, `metronome` clocks per click, 8 32nds
        data = (b'MThd' + struct.pack('>IHHH', 6, 1, 2, resolution) + self._encode_track(meta_track, end_of_track)
                + self._encode_track(self.track, end_of_track))
        with open(output_filename, 'wb') as f:
            f.write(data)
        return data


def write_sample(sample, outdir, fnm, isHalfAsSlow=False):
    if isHalfAsSlow:
        sample = np.repeat(sample, 2, axis=0)
    fnm = os.path.join(outdir, fnm + '.mid')
    MidiWriter().dump_sequence_to_midi(sample, fnm)
