"""Generates tests/golden/musical_metrics.npz by importing the REFERENCE's NumPy-only module
/root/reference/multinn/metrics/musical.py (the one reference module that runs in this container: SURVEY.md 8(c), 8(f) N2)
on seeded piano-rolls.  Run once here (the reference does not travel); the .npz holds inputs and the reference's outputs only."""
import importlib.util
import os
import sys

import numpy as np

REF = "/root/reference/multinn/metrics/musical.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_musical", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def rolls(seed, shape, rho, hold):
    """Seeded piano-rolls with held notes (runs of `hold`-ish steps) so that QN / PR / TD are not degenerate."""
    rng = np.random.Generator(np.random.PCG64(seed))
    B, bars, steps, P, M = shape
    T = bars * steps
    on = rng.random((B, T, P, M)) < rho
    x = np.zeros((B, T, P, M), bool)
    for d in range(hold):
        x[:, d:] |= on[:, :T - d] & (rng.random((B, T - d, P, M)) < 0.85 ** d)
    return x.reshape(B, bars, steps, P, M)


def main():
    m = load_reference()
    out = {}
    cases = [("a", 11, (3, 4, 16, 24, 5), 0.02, 4), ("b", 12, (2, 3, 48, 84, 5), 0.01, 6), ("c", 13, (4, 2, 96, 40, 5), 0.015, 8),
             ("d", 14, (2, 2, 24, 13, 5), 0.05, 3), ("e", 15, (1, 1, 32, 12, 5), 0.0, 1)]
    for name, seed, shape, rho, hold in cases:
        x = rolls(seed, shape, rho, hold)
        if name == "d":
            x[:, :, :, :, 2] = False                      # one silent track: EB = 1, QN = nan, TD rows nan
        chroma = m._to_chroma(x[..., 1:])
        out[f"{name}_x"] = np.packbits(x, axis=None)
        out[f"{name}_shape"] = np.array(shape)
        out[f"{name}_chroma"] = chroma
        out[f"{name}_eb"] = m.empty_bar_rate(x)
        out[f"{name}_up"] = m.num_pitches_used(x)
        out[f"{name}_upc"] = m.num_pitches_used(chroma)
        with np.errstate(all="ignore"):
            out[f"{name}_qn"] = np.array([m.qualified_note_rate(x[..., i:i + 1])[0] for i in range(shape[-1])])
            out[f"{name}_qn3"] = np.array([m.qualified_note_rate(x[..., i:i + 1], threshold=3)[0] for i in range(shape[-1])])
        out[f"{name}_pr"] = m.polyphonic_rate(x[..., 1:])
        out[f"{name}_pr1"] = m.polyphonic_rate(x, threshold=1)
        out[f"{name}_dp"] = np.float64(m.drum_in_pattern_rate(x[..., 0]))
        with np.errstate(all="ignore"), __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")
            out[f"{name}_td"] = m.harmonicity(chroma)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "musical_metrics.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
