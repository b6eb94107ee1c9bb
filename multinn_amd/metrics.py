"""metrics/statistical.py:6-47 -- per-batch scalar metrics (host-side reporting, off the hot path).

``batch/loss`` (statistical.py:34) is the optimised loss; the streaming tf.metrics become running
(total, count) pairs updated by calling the returned update closures.  R5 (f1 uses precision for
recall) is reproduced; R6 (perplexity = exp of a D-dim sum) is computed in float64 to avoid the
reference's float32 overflow.
"""
import torch


class Running:
    def __init__(self):
        self.total, self.count = 0.0, 0.0

    def update(self, total, count):
        self.total += float(total)
        self.count += float(count)

    def value(self):
        return self.total / self.count if self.count else 0.0


def base_metrics(loss, targets, predictions, log_probs):
    from . import ops             # one HIP pass over the cells (mnn_eval_counts); no CPU path: ops raises without a ROCm device
    cnt = torch.zeros(4, dtype=torch.int64, device=predictions.device)
    ops.eval_counts((targets.reshape(predictions.shape) != 0).to(torch.uint8).contiguous(), (predictions != 0).to(torch.uint8).contiguous(), cnt)
    tp, fp, fn, eq = (int(v) for v in cnt.tolist())
    n_cells = predictions.numel()
    state = {k: Running() for k in ("loss", "log_likelihood", "perplexity", "accuracy", "tp", "fp", "fn")}

    def upd():
        n = loss.numel()
        state["loss"].update(loss.sum(), n)
        state["log_likelihood"].update(log_probs.sum(), n)
        state["perplexity"].update(torch.exp(log_probs.double()).clamp(max=1e300).sum(), n)
        state["accuracy"].update(eq, n_cells)
        state["tp"].update(tp, 1); state["fp"].update(fp, 1); state["fn"].update(fn, 1)

    upd()

    def prec():
        d = state["tp"].total + state["fp"].total
        return state["tp"].total / d if d else 0.0

    def rec():
        d = state["tp"].total + state["fn"].total
        return state["tp"].total / d if d else 0.0

    metrics = {
        "loss": state["loss"].value(), "log_likelihood": state["log_likelihood"].value(),
        "perplexity": state["perplexity"].value(), "accuracy": state["accuracy"].value(),
        "precision": prec(), "recall": rec(),
        "batch/loss": loss.mean(),
    }
    pr = metrics["precision"]
    metrics["f1_score"] = 2 * (pr * pr) / (pr + pr) if pr > 0 else 0.0        # statistical.py:37-38 (R5)
    return metrics, [upd], None


# ------------------------------------------------------------------------------------------------
# Musical sample metrics (metrics/musical.py:16-275).  Same function names, arguments and return shapes
# as the reference (NumPy arrays out); the passes over the piano-roll run as two HIP kernels (mnn_musical_bar_stats,
# mnn_musical_note_stats), the few hundred resulting integers are turned into rates in float64 on the host.
# Inputs: torch tensors (any device; moved to the ROCm device) or NumPy arrays, `[batch, bars, steps, pitch, tracks]`,
# any non-zero cell = note on.
# ------------------------------------------------------------------------------------------------
import ctypes as _C

import numpy as _np

_PATTERNS = {96: ((1, 2, 0, 0, 0, 2), 16), 48: ((1, 2, 2), 16), 24: ((1, 2, 2), 8), 72: ((1, 2, 0, 0, 0, 2), 12), 36: ((1, 2, 2), 12),
             64: ((1, 2, 0, 2), 16), 32: ((1, 2), 16), 16: ((1, 2), 8)}       # musical.py:148-168 (1 = weight 1, 2 = weight `tolerance`)


def _as_u8(x, ndim):
    from .common import default_device
    t = torch.as_tensor(x)
    if t.dim() != ndim:
        raise ValueError(f"Input tensor must have {ndim} dimensions.")       # musical.py:54-55
    return (t != 0).to(device=default_device(), dtype=torch.uint8).contiguous()


def _bar_stats(x5, poly_threshold=2, pattern=False):
    """One pass of mnn_musical_bar_stats over u8 [B,bars,steps,P,M] -> dict of int64 NumPy tables [B,bars,M] (+ beat chroma)."""
    from . import ops
    B, bars, steps, P, M = x5.shape
    dev = x5.device
    pat = None
    if pattern:
        if steps not in _PATTERNS:
            raise ValueError("Unsupported number of timesteps for the drum in pattern metric.")      # musical.py:167-168
        p, rep = _PATTERNS[steps]
        pat = torch.tensor(list(p) * rep, dtype=torch.uint8, device=dev)
    out = {k: torch.empty((B * bars, M), dtype=torch.int32, device=dev) for k in ("notes", "used_pitches", "used_classes", "poly_steps", "pat_on", "pat_tol")}
    bc = torch.empty((B * bars, 4, 12, M), dtype=torch.int32, device=dev)
    ops.musical_bar_stats(x5, poly_threshold, pat, out["notes"], out["used_pitches"], out["used_classes"], out["poly_steps"], out["pat_on"],
                          out["pat_tol"], bc)
    res = {k: v.view(B, bars, M).cpu().numpy().astype(_np.int64) for k, v in out.items()}
    res["beat_chroma"] = bc.view(B * bars * 4, 12, M).cpu().numpy().astype(_np.int64)
    return res


def _to_chroma(pianoroll):
    """musical.py:16-41: pitches zero-padded to a multiple of 12 and block-folded (class = p // (P_padded/12)) -> counts
    `[batch, bars, steps, 12, tracks]` (a reshape + sum: tensor plumbing, kept on the device)."""
    t = torch.as_tensor(pianoroll)
    P = t.shape[-2]
    per = -(-P // 12)
    t = (t != 0).to(torch.int32)
    if per * 12 != P:
        t = torch.nn.functional.pad(t, (0, 0, 0, per * 12 - P))
    return t.reshape(t.shape[:-2] + (12, per, t.shape[-1])).sum(dim=-2)


def empty_bar_rate(pianoroll):
    """musical.py:45-57."""
    s = _bar_stats(_as_u8(pianoroll, 5))
    return 1 - (s["notes"] > 0).mean(axis=(0, 1))


def num_pitches_used(pianoroll):
    """musical.py:60-73 (also applied to chroma tensors, :250-251)."""
    s = _bar_stats(_as_u8(pianoroll, 5))
    return s["used_pitches"].mean(axis=(0, 1))


def qualified_note_rate(pianoroll, threshold=2):
    """musical.py:76-113.  The reference's denominator is count_nonzero of the onsets' flat positions (:108-111), so a note that
    starts at sample 0, pitch 0, step 0 is not counted there; reproduced."""
    from . import ops
    x = _as_u8(pianoroll, 5)
    B, bars, steps, P, M = x.shape
    on = torch.zeros(M, dtype=torch.int32, device=x.device)
    q = torch.zeros(M, dtype=torch.int32, device=x.device)
    ops.musical_note_stats(x.view(B, bars * steps, P, M), threshold, on, q)
    n_on = on.cpu().numpy().astype(_np.float32) - x[0, 0, 0, 0, :].cpu().numpy().astype(_np.float32)
    with _np.errstate(divide="ignore", invalid="ignore"):
        return q.cpu().numpy().astype(_np.float32) / n_on


def polyphonic_rate(pianoroll, threshold=2):
    """musical.py:116-132 (steps with MORE than `threshold` pitches)."""
    x = _as_u8(pianoroll, 5)
    s = _bar_stats(x, poly_threshold=threshold)
    return (s["poly_steps"] / x.shape[2]).mean(axis=(0, 1))


def drum_in_pattern_rate(chroma, tolerance=0.1):
    """musical.py:135-178; `chroma` is the drum track `[batch, bars, steps, pitch]` as the reference passes it (:267)."""
    x = _as_u8(chroma, 4).unsqueeze(-1)
    s = _bar_stats(x, pattern=True)
    notes = int(s["notes"].sum())
    num = float(s["pat_on"].sum()) + tolerance * float(s["pat_tol"].sum())
    return num / notes if notes > 0 else 0.


def _tonal_matrix(r1=1.0, r2=1.0, r3=0.5):
    k = _np.arange(12)                                                        # musical.py:202-215 (Harte et al. 2006)
    return _np.stack([r1 * _np.sin(k * 7. / 6. * _np.pi), r1 * _np.cos(k * 7. / 6. * _np.pi), r2 * _np.sin(k * 3. / 2. * _np.pi),
                      r2 * _np.cos(k * 3. / 2. * _np.pi), r3 * _np.sin(k * 2. / 3. * _np.pi), r3 * _np.cos(k * 2. / 3. * _np.pi)])


def _tonal_distance(beat):
    """beat int [nb, 12, M] (notes per beat and chroma class) -> [M, M] (musical.py:217-236)."""
    import warnings
    beat = beat.astype(_np.float64)
    with _np.errstate(divide="ignore", invalid="ignore"):
        beat = beat / beat.sum(axis=1, keepdims=True)
    pts = _np.einsum("kc,ncm->knm", _tonal_matrix(), beat)
    dist = _np.sqrt(((pts[:, :, :, None] - pts[:, :, None, :]) ** 2).sum(axis=0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _np.nanmean(dist, axis=0)


def harmonicity(chroma):
    """musical.py:181-236: chroma `[batch, bars, steps, 12, tracks]` (counts) -> track-to-track tonal distance matrix."""
    t = torch.as_tensor(chroma)
    if t.dim() != 5:
        raise ValueError("Input tensor must have 5 dimensions.")
    if t.shape[3] != 12:
        raise ValueError("Input tensor must be a chroma tensor.")
    B, bars, steps, _, M = t.shape
    beat = t.reshape(B * bars * 4, steps // 4, 12, M).sum(dim=1).cpu().numpy()
    return _tonal_distance(beat)


def sample_metrics(pianoroll):
    """All of compute_sample_metrics' numbers (musical.py:239-275) as a dict, from ONE bar pass over the full roll plus one
    note pass: EB, UP over all tracks; UPC, QN, PR, TD over tracks 1..M-1; DP on track 0."""
    from . import ops
    x = _as_u8(pianoroll, 5)
    B, bars, steps, P, M = x.shape
    s = _bar_stats(x, poly_threshold=2, pattern=steps in _PATTERNS)
    on = torch.zeros(M, dtype=torch.int32, device=x.device)
    q = torch.zeros(M, dtype=torch.int32, device=x.device)
    ops.musical_note_stats(x.view(B, bars * steps, P, M), 2, on, q)
    on, q = on.cpu().numpy().astype(_np.float32), q.cpu().numpy().astype(_np.float32)
    on[0] -= float(x[0, 0, 0, 0, 0])          # only a single-track call starts its flat view at (sample 0, pitch 0) of that track ...
    # ... the reference evaluates QN per track slice pianoroll[..., i:i+1] (:256-258), so every track has that quirk:
    on[1:] -= x[0, 0, 0, 0, 1:].cpu().numpy().astype(_np.float32)
    with _np.errstate(divide="ignore", invalid="ignore"):
        qn = q / on
    # chroma of tracks 1.. uses the class fold of THEIR pitch axis (same P): used_classes / beat_chroma columns 1..
    dp = 0.
    if steps in _PATTERNS:
        notes0 = int(s["notes"][..., 0].sum())
        dp = (float(s["pat_on"][..., 0].sum()) + 0.1 * float(s["pat_tol"][..., 0].sum())) / notes0 if notes0 > 0 else 0.
    return {"EB": 1 - (s["notes"] > 0).mean(axis=(0, 1)), "UP": s["used_pitches"].mean(axis=(0, 1)), "UPC": s["used_classes"][..., 1:].mean(axis=(0, 1)),
            "QN": qn[1:], "PR": (s["poly_steps"][..., 1:] / steps).mean(axis=(0, 1)), "DP": dp, "TD": _tonal_distance(s["beat_chroma"][..., 1:])}


def compute_sample_metrics(pianoroll):
    """musical.py:239-275: evaluates sample bar piano-rolls and prints the table the reference prints."""
    m = sample_metrics(pianoroll)
    fmt = lambda v: '  '.join(f'{x:.5f}' for x in v)
    print()
    print(' ' * 5 + ' Drums    Piano    Guitar   Bass    Strings')
    print(f'{"EB: ":5s}' + fmt(m["EB"]))
    print(f'{"UP: ":5s}' + fmt(m["UP"]))
    print(f'{"UPC: ":5s}   -     ' + fmt(m["UPC"]))
    print(f'{"QN: ":5s}   -     ' + '  '.join(f'{x:.5f}' for x in m["QN"]))
    print(f'\n{"PR: ":5s}   -     ' + fmt(m["PR"]))
    print(f'\n{"DP: ":5s}{m["DP"]:.5f}')
    print(f'\n{"TD: ":5s}')
    print(m["TD"])
    return m
