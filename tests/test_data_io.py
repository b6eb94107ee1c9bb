"""N3 (utils/data.py:8-202): dataset / sampling / MIDI formats either side of the path.  Host-side NumPy; parity unpinned (the
reference module needs pypianoroll): checked by shape / value properties stated in the reference's docstrings and by reading the
written MIDI file back."""
import os
import struct

import numpy as np
import pytest

from multinn_amd import data as D

CFG = {'source': 'npy', 'filename': None, 'sequence_lengths': None, 'beat_resolution': 12, 'pitch_range': {'lowest': 24, 'highest': 108},
       'instruments': ['Drums', 'Piano', 'Guitar ', 'Bass', 'Strings'], 'programs': [0, 0, 24, 32, 48], 'is_drums': [True, False, False, False, False],
       'tempo': 120, 'split': {'num_train': 6, 'num_valid': 2, 'num_test': 2}}


def test_load_data_reshape_split_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    songs = (rng.random((12, 50, 84, 5)) < 0.05)
    cfg = dict(CFG, filename=str(tmp_path / 'X'))
    np.save(cfg['filename'] + '.npy', songs)
    (tr, ltr), (va, lva), (te, lte) = D.load_data(cfg, step_size=3)          # 50 -> padded to 51 pixels -> 17 steps of 3 x 84
    assert tr.shape == (6, 17, 252, 5) and va.shape == (2, 17, 252, 5) and te.shape == (2, 17, 252, 5)
    assert (ltr == 17).all() and len(lva) == 2 and len(lte) == 2
    assert np.array_equal(tr[3, 4].reshape(3, 84, 5), songs[3, 12:15])       # step s = pixels 3s .. 3s+2
    assert not tr[:, 16].reshape(6, 3, 84, 5)[:, 2].any()                    # the padded pixel
    assert np.array_equal(te, D.load_data(cfg, 3)[2][0]) and np.array_equal(te[0].reshape(51, 84, 5)[:50], songs[8])   # test = LAST songs of the first 10
    np.save(str(tmp_path / 'L.npy'), np.arange(12))
    assert list(D.load_data(dict(cfg, sequence_lengths=str(tmp_path / 'L.npy')), 1)[1][1]) == [6, 7]
    with pytest.raises(ValueError):
        D.load_data(dict(cfg, source='dir'))
    np.save(cfg['filename'] + '.npy', songs[..., :4])
    with pytest.raises(ValueError):
        D.load_data(cfg)
    np.save(cfg['filename'] + '.npy', songs[0])
    with pytest.raises(ValueError):
        D.load_data(cfg)


def test_pad_to_midi_and_sampling_inputs():
    x = np.ones((2, 4, 252, 5), np.uint8)                                     # 4 steps of 3 pixels x 84 pitches
    p = D.pad_to_midi(x, CFG)
    assert p.shape == (2, 12, 128, 5) and p[:, :, :24].sum() == 0 and p[:, :, 108:].sum() == 0 and p[:, :, 24:108].all()
    Xtr, Xva = np.arange(10 * 20 * 3 * 2).reshape(10, 20, 3, 2), -np.arange(6 * 20 * 3 * 2).reshape(6, 20, 3, 2)
    sc = {'intro_beats': 2, 'intro_ids': {'train': {'start': 1, 'end': 4}, 'valid': {'start': 0, 'end': 2}},
          'save_ids': {'train': [0, 2], 'valid': [1]}, 'num_save': 3}
    intro, ids, labels = D.prepare_sampling_inputs(Xtr, Xva, sc, beat_size=4)
    assert intro.shape == (5, 8, 3, 2) and np.array_equal(intro[:3], Xtr[1:4, :8]) and np.array_equal(intro[3:], Xva[0:2, :8])
    assert labels == ['t0', 't2', 'v1'] and list(ids) == [0, 2, 4, 5, 7, 9, 10, 12, 14]     # valid ids offset by the 3 train intros, then +5 per extra sample


def _read_smf(path):
    b = open(path, 'rb').read()
    assert b[:4] == b'MThd'
    fmt, ntr, div = struct.unpack('>HHH', b[8:14])
    pos, tracks = 14, []
    for _ in range(ntr):
        assert b[pos:pos + 4] == b'MTrk'
        n = struct.unpack('>I', b[pos + 4:pos + 8])[0]
        d, q, tick, ev, status = b[pos + 8:pos + 8 + n], 0, 0, [], None
        while q < len(d):
            dt = 0
            while True:
                dt = (dt << 7) | (d[q] & 0x7F); q += 1
                if not d[q - 1] & 0x80:
                    break
            tick += dt
            if d[q] == 0xFF:
                ln = d[q + 2]; ev.append((tick, 'meta', d[q + 1], bytes(d[q + 3:q + 3 + ln]))); q += 3 + ln
            else:
                status = d[q]; k = 2 if (status & 0xF0) == 0xC0 else 3
                ev.append((tick, status, *d[q + 1:q + k])); q += k
        tracks.append(ev); pos += 8 + n
    return fmt, div, tracks


def test_write_song_round_trip(tmp_path):
    T = 48
    song = np.zeros((T, 128, 5), np.uint8)
    song[0:12, 36, 0] = 1; song[24:25, 38, 0] = 1                              # drums
    song[0:24, 60, 1] = 1; song[24:48, 64, 1] = 1                              # piano: two half-bar notes
    song[6:18, 40, 3] = 1                                                      # bass
    D.save_music(D.pad_to_midi(song[None, :, 24:108], CFG), 1, CFG, 'demo', save_dir=str(tmp_path), song_labels=['t0'])
    fmt, div, tracks = _read_smf(os.path.join(tmp_path, 'demo_t0_0.mid'))
    assert fmt == 1 and div == 12 and len(tracks) == 6
    assert tracks[0][0][:3] == (0, 'meta', 0x51) and int.from_bytes(tracks[0][0][3], 'big') == 500000          # 120 bpm
    notes = lambda tr: [(e[0], e[1] & 0xF0, e[1] & 0x0F, e[2], e[3]) for e in tr if e[1] != 'meta' and (e[1] & 0xF0) in (0x80, 0x90)]
    assert notes(tracks[1]) == [(0, 0x90, 9, 36, 100), (12, 0x80, 9, 36, 0), (24, 0x90, 9, 38, 100), (25, 0x80, 9, 38, 0)]      # drums on channel 10
    assert notes(tracks[2]) == [(0, 0x90, 0, 60, 80), (24, 0x80, 0, 60, 0), (24, 0x90, 0, 64, 80), (48, 0x80, 0, 64, 0)]       # piano gain 0.8
    assert notes(tracks[4]) == [(6, 0x90, 2, 40, 120), (18, 0x80, 2, 40, 0)]                                                    # bass gain 1.2
    assert [e for e in tracks[3] if e[1] != 'meta' and (e[1] & 0xF0) == 0xC0][0][2] == 24 and notes(tracks[3]) == []            # guitar: program only
