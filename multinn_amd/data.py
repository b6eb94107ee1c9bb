"""utils/data.py:8-202 -- the on-disk formats either side of the hot path: the `.npy` dataset loader with the
`num_pixels` time-step reshape, intro selection for sampling, padding back to 128 MIDI pitches, and MIDI output.

Host-side NumPy only (no device work).  The reference writes MIDI through `pypianoroll` (absent here and not a dependency of this
package); `write_song` emits the same piano-rolls as a Standard MIDI File (format 1) with a small writer of its own: one tempo
track + one track per instrument, `beat_resolution` ticks per quarter note, a note per maximal run of non-zero cells of a pitch,
velocity = the run's first cell (the reference scales the binary roll by 100 and by a per-instrument gain, data.py:153-166).
"""
import os
import struct

import numpy as np


def pad_to_midi(songs, data_config):
    """data.py:8-35: `[batch, time_steps, step_span, tracks]` -> `[batch, steps, 128, tracks]` (zero rows below / above the range)."""
    lo, hi = data_config['pitch_range']['lowest'], data_config['pitch_range']['highest']
    songs = np.reshape(songs, (songs.shape[0], -1, hi - lo, songs.shape[-1]))
    return np.pad(songs, ((0, 0), (0, 0), (lo, 128 - hi), (0, 0)), 'constant', constant_values=0)


def load_data(data_config, step_size=1):
    """data.py:38-100: loads `<filename>.npy` `[songs, pixels, pitches, tracks]`, zero-pads the time axis to a multiple of
    `step_size` and folds `step_size` pixels into one model time step (`[songs, pixels/step, pitches*step, tracks]`); returns the
    (data, lengths) pairs of the train / validation / test split (test = the LAST num_test songs, as written)."""
    path = data_config['filename']
    n_tr, n_va, n_te = (data_config['split'][k] for k in ('num_train', 'num_valid', 'num_test'))
    if data_config['source'] != 'npy':
        raise ValueError('Not supported data format :(')
    songs = np.load(f'{path}.npy')[:n_tr + n_va + n_te]
    if len(songs.shape) != 4:
        raise ValueError("Dataset must have 4 dimensions.")
    if songs.shape[-1] != len(data_config['instruments']):
        raise ValueError(f"Dataset must have {len(data_config['instruments'])} tracks.")
    pad = (-songs.shape[1]) % step_size
    if pad:
        songs = np.pad(songs, ((0, 0), (0, pad), (0, 0), (0, 0)), 'constant', constant_values=0)
    songs = songs.reshape([songs.shape[0], songs.shape[1] // step_size, songs.shape[2] * step_size, songs.shape[3]])
    if data_config.get('sequence_lengths'):
        lengths = np.load(data_config['sequence_lengths'])[:n_tr + n_va + n_te]
    else:
        lengths = np.full(songs.shape[0], songs.shape[1])
    return ((songs[:n_tr], lengths[:n_tr]), (songs[n_tr:n_tr + n_va], lengths[n_tr:n_tr + n_va]), (songs[-n_te:], lengths[-n_te:]))


def prepare_sampling_inputs(X_train, X_valid, sampling_config, beat_size):
    """data.py:103-141: the intro excerpts (first `intro_beats` beats of the configured train / validation songs), the ids of the
    samples to save (`num_save` samples per intro: ids repeat with stride len(intro_songs)) and their labels t<i> / v<i>."""
    intro_steps = int(sampling_config['intro_beats'] * beat_size)
    ids = sampling_config['intro_ids']
    intro_songs = np.concatenate([X_train[ids['train']['start']:ids['train']['end'], :intro_steps, :],
                                  X_valid[ids['valid']['start']:ids['valid']['end'], :intro_steps, :]], axis=0)
    save_train = np.array(sampling_config['save_ids']['train'])
    save_valid = np.array(sampling_config['save_ids']['valid'])
    song_labels = [f't{i}' for i in save_train] + [f'v{i}' for i in save_valid]
    save_ids = np.concatenate([save_train, save_valid + (ids['train']['end'] - ids['train']['start'])], axis=0)
    nxt = save_ids
    for _ in range(1, sampling_config['num_save']):
        nxt = nxt + len(intro_songs)
        save_ids = np.concatenate([save_ids, nxt], axis=0)
    return intro_songs, save_ids, song_labels


# ---- Standard MIDI File writer ---------------------------------------------------------------------------------------
def _vlq(n):
    out = [n & 0x7F]
    n >>= 7
    while n:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    return bytes(reversed(out))


def _track_chunk(events):
    """events: (absolute tick, order, bytes); sorted by tick (note-offs before note-ons at the same tick)."""
    data, last = bytearray(), 0
    for tick, _, ev in sorted(events, key=lambda e: (e[0], e[1])):
        data += _vlq(tick - last) + ev
        last = tick
    data += b'\x00\xff\x2f\x00'
    return b'MTrk' + struct.pack('>I', len(data)) + bytes(data)


_GAIN = {'Piano': 0.8, 'Strings': 0.9, 'Bass': 1.2}           # data.py:158-164


def write_song(song, path, data_config):
    """data.py:144-181: `[time_steps, 128, tracks]` piano-roll -> MIDI file at `path`."""
    song = np.asarray(song, np.float64) * 100.
    names = data_config['instruments']
    res = int(data_config['beat_resolution'])
    tempo = int(round(60_000_000 / float(data_config['tempo'])))
    chunks = [_track_chunk([(0, 0, b'\xff\x51\x03' + struct.pack('>I', tempo)[1:])])]
    melodic = 0
    for i, name in enumerate(names):
        roll = np.take(song, i, axis=-1) * _GAIN.get(name.strip(), 1.0)
        drum = bool(data_config['is_drums'][i])
        if drum:
            ch = 9
        else:
            ch = melodic if melodic < 9 else melodic + 1
            melodic += 1
        ev = [(0, 0, b'\xff\x03' + _vlq(len(name.strip())) + name.strip().encode()),
              (0, 1, bytes([0xC0 | ch, int(data_config['programs'][i]) & 0x7F]))]
        on = roll > 0
        edge = np.diff(np.concatenate([np.zeros((1, roll.shape[1]), bool), on, np.zeros((1, roll.shape[1]), bool)]).astype(np.int8), axis=0)
        for pitch in np.nonzero(on.any(axis=0))[0]:
            starts, ends = np.nonzero(edge[:, pitch] > 0)[0], np.nonzero(edge[:, pitch] < 0)[0]
            for s, e in zip(starts, ends):
                vel = int(min(127, max(1, round(float(roll[s, pitch])))))
                ev.append((int(s), 3, bytes([0x90 | ch, int(pitch), vel])))
                ev.append((int(e), 2, bytes([0x80 | ch, int(pitch), 0])))
        chunks.append(_track_chunk(ev))
    with open(path, 'wb') as f:
        f.write(b'MThd' + struct.pack('>IHHH', 6, 1, len(chunks), res) + b''.join(chunks))


def save_music(music, num_intro, data_config, base_path, save_dir='outputs/', song_labels=None):
    """data.py:184-202: `[num_songs * num_intro, steps, 128, tracks]` -> `<base>_<label>_<j>.mid` for sample j of intro i."""
    os.makedirs(save_dir, exist_ok=True)
    for i in range(num_intro):
        for j in range(music.shape[0] // num_intro):
            label = f'song{i}' if song_labels is None else song_labels[i]
            write_song(music[i + j * num_intro], os.path.join(save_dir, f'{base_path}_{label}_{j}.mid'), data_config)
