"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's musical sample metrics
(/root/reference/multinn/metrics/musical.py:16-275) in NumPy, written from the metric definitions.

PARITY PINNED for this component: tests/golden/musical_metrics.npz holds outputs of the REFERENCE module itself
(generated in this container by tests/golden/make_musical_golden.py; the module is NumPy-only and runs here), and
tests/test_oracle_kats.py checks every function below against them.

Shapes: piano-roll `[batch, bars, steps, pitch, tracks]` (bool / 0-1), chroma `[batch, bars, steps, 12, tracks]` (counts).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import warnings

import numpy as np


def to_chroma(x):
    """musical.py:16-41.  Pitches are zero-padded up to a multiple of 12 and folded as reshape(12, P/12).sum over the second
    factor: class(p) = p // (P_padded / 12) -- a block fold, not p % 12."""
    P = x.shape[-2]
    per = -(-P // 12)
    cls = np.arange(P) // per
    out = np.zeros(x.shape[:-2] + (12, x.shape[-1]), np.int64)
    for c in range(12):
        sel = cls == c
        if sel.any():
            out[..., c, :] = x[..., sel, :].sum(axis=-2)
    return out


def _need(x, n):
    if x.ndim != n:
        raise ValueError(f"Input tensor must have {n} dimensions.")          # musical.py:54-55 etc.


def empty_bar_rate(x):
    """musical.py:45-57: share of (sample, bar) pairs without any note, per track."""
    _need(x, 5)
    nonempty = x.astype(bool).any(axis=(2, 3))                   # [B, bars, M]
    return 1 - nonempty.mean(axis=(0, 1))


def num_pitches_used(x):
    """musical.py:60-73: mean over (sample, bar) of the number of pitches (or chroma classes) that sound in the bar."""
    _need(x, 5)
    used = (x != 0).any(axis=2)                                  # [B, bars, P, M]
    return used.sum(axis=2).mean(axis=(0, 1))


def note_runs(x):
    """Maximal runs along time of each (sample, pitch, track), bars concatenated (musical.py:93-96) -> (lengths, track, first)
    where `first` marks the run that starts at flat position 0 of the reference's [track, sample*pitch*(T+1)] view."""
    B, bars, steps, P, M = x.shape
    r = x.reshape(B, bars * steps, P, M).astype(bool)
    lens, trk, first = [], [], []
    for m in range(M):
        for b in range(B):
            for p in range(P):
                col = r[b, :, p, m]
                d = np.diff(np.concatenate(([0], col.view(np.int8), [0])))
                on, off = np.nonzero(d > 0)[0], np.nonzero(d < 0)[0]
                lens.extend((off - on).tolist()); trk.extend([m] * len(on))
                first.extend([(b == 0 and p == 0 and o == 0) for o in on])
    return np.array(lens, np.int64), np.array(trk, np.int64), np.array(first, bool)


def qualified_note_rate(x, threshold=2):
    """musical.py:76-113: notes LONGER than `threshold` steps over the number of onsets, per track.  As written, the
    denominator is count_nonzero of the onsets' flat POSITIONS (:108-111), so an onset at flat position 0 -- sample 0,
    pitch 0, step 0 -- is not counted; reproduced."""
    _need(x, 5)
    lens, trk, first = note_runs(x)
    M = x.shape[-1]
    q = np.array([np.count_nonzero(lens[trk == m] > threshold) for m in range(M)], np.float32)
    n = np.array([np.count_nonzero(trk == m) - np.count_nonzero(first[trk == m]) for m in range(M)], np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return q / n


def polyphonic_rate(x, threshold=2):
    """musical.py:116-132: share of steps with MORE than `threshold` pitches on, averaged over (sample, bar), per track."""
    _need(x, 5)
    per_step = (x != 0).sum(axis=3)                              # [B, bars, steps, M]
    return ((per_step > threshold).sum(axis=2) / x.shape[2]).mean(axis=(0, 1))


def drum_pattern_mask(steps, tolerance=0.1):
    """musical.py:148-168: the table of supported bar resolutions."""
    table = {96: ([1., tolerance, 0., 0., 0., tolerance], 16), 48: ([1., tolerance, tolerance], 16), 24: ([1., tolerance, tolerance], 8),
             72: ([1., tolerance, 0., 0., 0., tolerance], 12), 36: ([1., tolerance, tolerance], 12), 64: ([1., tolerance, 0., tolerance], 16),
             32: ([1., tolerance], 16), 16: ([1., tolerance], 8)}
    if steps not in table:
        raise ValueError("Unsupported number of timesteps for the drum in pattern metric.")
    pat, rep = table[steps]
    return np.tile(pat, rep)


def drum_in_pattern_rate(d):
    """musical.py:135-178: d `[batch, bars, steps, pitch]` (the drum track as it is passed at :267): mask-weighted note count
    over the note count; 0 when there is no note."""
    _need(d, 4)
    mask = drum_pattern_mask(d.shape[2])
    per_step = d.astype(np.float64).sum(axis=3)                  # [B, bars, steps]
    num = float((per_step * mask.reshape(1, 1, -1)).sum())
    notes = int(np.count_nonzero(d))
    return num / notes if notes > 0 else 0.


def tonal_matrix(r1=1.0, r2=1.0, r3=0.5):
    """musical.py:202-215 (Harte et al. 2006): rows sin/cos of the fifths, minor-thirds and major-thirds circles."""
    k = np.arange(12)
    return np.stack([r1 * np.sin(k * 7. / 6. * np.pi), r1 * np.cos(k * 7. / 6. * np.pi), r2 * np.sin(k * 3. / 2. * np.pi),
                     r2 * np.cos(k * 3. / 2. * np.pi), r3 * np.sin(k * 2. / 3. * np.pi), r3 * np.cos(k * 2. / 3. * np.pi)])


def harmonicity(chroma):
    """musical.py:181-236: chroma summed per beat (a quarter of the bar) and normalised over the 12 classes, mapped to the
    6-D tonal space; the Euclidean distance between every pair of tracks, nan-averaged over the beats."""
    _need(chroma, 5)
    if chroma.shape[3] != 12:
        raise ValueError("Input tensor must be a chroma tensor.")
    B, bars, steps, _, M = chroma.shape
    beat = chroma.reshape(B * bars * 4, steps // 4, 12, M).sum(axis=1).astype(np.float64)      # [nb, 12, M]
    with np.errstate(divide="ignore", invalid="ignore"):
        beat = beat / beat.sum(axis=1, keepdims=True)
    pts = np.einsum("kc,ncm->knm", tonal_matrix(), beat)          # [6, nb, M]
    dist = np.sqrt(((pts[:, :, :, None] - pts[:, :, None, :]) ** 2).sum(axis=0))               # [nb, M, M]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return np.nanmean(dist, axis=0)
