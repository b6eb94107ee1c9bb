"""N3 through the device path: the on-disk formats either side of the hot path exercised as the reference's drivers use them --
`.npy` dataset -> `load_data` (num_pixels fold, split) -> the joint LSTM-NADE mode trains on the loaded windows (train.py:142-194) -> intros by
`prepare_sampling_inputs` -> `generate` (sample.py:52) -> `pad_to_midi` -> `save_music` -> the written Standard MIDI Files read back note for
note.  (No reference-held fixture exists for utils/data.py: it needs pypianoroll; the bytes are checked against the generated piano-roll.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_data_io import _read_smf   # noqa: E402  (the SMF reader of the CPU test)

DEV = "cuda:0"


def test_dataset_to_midi_pipeline_through_the_joint_mode(tmp_path):
    from multinn_amd import data as D, MultINN, AdamOptimizer
    lo, hi, M, res = 40, 52, 3, 4                       # 12 pitches, 3 tracks, 4 pixels per beat
    names = ['Drums', 'Piano', 'Bass']
    rng = np.random.default_rng(5)
    songs = (rng.random((10, 64, hi - lo, M)) < 0.12).astype(np.uint8)        # [songs, pixels, pitches, tracks] as prepare_data.py:28 stores them
    np.save(tmp_path / "set.npy", songs)
    dcfg = {'source': 'npy', 'filename': str(tmp_path / "set"), 'sequence_lengths': None, 'beat_resolution': res,
            'pitch_range': {'lowest': lo, 'highest': hi}, 'instruments': names, 'programs': [0, 0, 32], 'is_drums': [True, False, False],
            'tempo': 100, 'split': {'num_train': 6, 'num_valid': 2, 'num_test': 2}}
    num_pixels = 2
    (Xtr, Ltr), (Xva, Lva), (Xte, Lte) = D.load_data(dcfg, step_size=num_pixels)
    assert Xtr.shape == (6, 32, 24, M) and Xva.shape == (2, 32, 24, M) and Xte.shape == (2, 32, 24, M) and (Ltr == 32).all()
    config = {"model_name": "pipe", "data": dcfg, "training": {"num_pixels": num_pixels, "random_seed": 23}}
    params = {"mode": "joint", "tune_encoder": False, "keep_prob": 0.9, "encoder": {"type": "Pass", "num_hidden": None},
              "generator": {"type": "NADE", "num_hidden": 16, "num_hidden_rnn": [32, 32], "feedback": None}}
    model = MultINN(config, params, mode="joint", precision="fp16")
    opt = AdamOptimizer(0.01)
    x = torch.from_numpy(Xtr).to(DEV)
    losses = [float(model.train_step(x[:, t0:t0 + 16], None, opt)) for _ in range(6) for t0 in (0, 16)]       # piece_size 16 windows
    model.check()
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    beat_size = res // num_pixels                        # model time steps per beat
    scfg = {'intro_beats': 4, 'num_save': 2, 'intro_ids': {'train': {'start': 0, 'end': 2}, 'valid': {'start': 0, 'end': 1}},
            'save_ids': {'train': [0, 1], 'valid': [0]}}
    intros, save_ids, labels = D.prepare_sampling_inputs(Xtr, Xva, scfg, beat_size)
    assert intros.shape == (3, 8, 24, M) and labels == ['t0', 't1', 'v0'] and save_ids.tolist() == [0, 1, 2, 3, 4, 5]
    rep = np.concatenate([intros] * scfg['num_save'], axis=0)                  # sample.py:45-47: every intro num_save times
    model.build(torch.from_numpy(rep).to(DEV), lengths=None, is_train=False, mode="generate")
    gen = model.generate(12).cpu().numpy()                                     # [6, 12, 24, M]
    assert gen.shape == (6, 12, 24, M) and gen.dtype == np.uint8 and 0 < gen.mean() < 1
    music = D.pad_to_midi(np.concatenate([rep, gen], axis=1), dcfg)            # intro + continuation, back to pixels x 128 pitches
    assert music.shape == (6, 40, 128, M) and not music[:, :, :lo].any() and not music[:, :, hi:].any()
    D.save_music(music[save_ids], len(labels), dcfg, "pipe", save_dir=str(tmp_path / "out"), song_labels=labels)
    files = sorted(os.listdir(tmp_path / "out"))
    assert files == [f"pipe_{l}_{j}.mid" for l in labels for j in range(2)]
    for i, lab in enumerate(labels):
        for j in range(2):
            fmt, div, tracks = _read_smf(os.path.join(tmp_path, "out", f"pipe_{lab}_{j}.mid"))
            assert fmt == 1 and div == res and len(tracks) == M + 1
            roll = music[save_ids][i + j * len(labels)]                        # [40, 128, M]
            for m in range(M):
                on = {}
                rebuilt = np.zeros((40, 128), bool)
                for e in tracks[m + 1]:
                    if e[1] == 'meta' or (e[1] & 0xF0) not in (0x80, 0x90):
                        continue
                    if (e[1] & 0xF0) == 0x90 and e[3] > 0:
                        on[e[2]] = e[0]
                    else:
                        rebuilt[on.pop(e[2]):e[0], e[2]] = True
                assert not on and np.array_equal(rebuilt, roll[:, :, m] > 0), (lab, j, m)
