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
    def __init__(self, verbose=False, default_vel=100):
        self.verbose = verbose
        self.note_range = RANGE
        self.default_velocity = default_vel

    def note_off(self, val, tick):
        self.track.append((tick, 0x80, (val, 0)))
        return 0

    def note_on(self, val, tick):
        self.track.append((tick, 0x90, (val, self.default_velocity)))
        return 0

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

    def dump_sequence_to_midi(self, seq, output_filename, time_step=120, resolution=480, metronome=24, offset=21,
                              format='final', end_of_track=False):
        if format == 'icml':
            sequence = np.array([[1 if i in tm else 0 for i in range(self.note_range)] for tm in seq])
        elif format == 'flat':
            sequence = np.reshape(seq, [-1, self.note_range])
        else:
            sequence = np.asarray(seq)
        self.track = []
        meta_track = [(0, 0xFF, (0x58, 4, 2, metronome, 8))]
        tick = time_step
        self.notes_on = {n: False for n in range(self.note_range)}
        for seq_idx in range(sequence.shape[0]):
            notes = [n + offset for n in np.nonzero(sequence[seq_idx, :])[0].tolist()]
            for n in self.notes_on:
                if self.notes_on[n] and n not in notes:
                    tick = self.note_off(n, tick)
                    self.notes_on[n] = False
            for note in notes:
                if not self.notes_on[note]:
                    tick = self.note_on(note, tick)
                    self.notes_on[note] = True
            tick += time_step
        for n in self.notes_on:
            if self.notes_on[n]:
                self.note_off(n, tick)
                tick = 0
                self.notes_on[n] = False
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
