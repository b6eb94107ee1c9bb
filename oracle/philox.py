"""Philox4x32-10 counter-based RNG and the TF float32 uniform mapping (oracle, NumPy).

TEST INFRASTRUCTURE ONLY.  Algorithm: Salmon et al., "Parallel Random Numbers: As
Easy as 1, 2, 3" (SC'11), the generator TF's ``random_uniform`` is built on
(SURVEY.md 8(c) item 5).  The (key, counter) assignment is this build's own RNG
contract (DESIGN.md "RNG contract"):

    key     = (seed_lo, seed_hi)
    counter = (elem >> 2, row, sub, stream)        # 4 x uint32
    word    = elem & 3                             # which of the 4 outputs
    u       = bitcast_f32((x & 0x7fffff) | 0x3f800000) - 1.0      # 23-bit grid in [0,1)

Streams: 0 dropout, 1 NADE sample, 2 RBM hidden, 3 RBM visible, 4 DBN encode,
5 DBN decode.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_DROPOUT, STREAM_NADE, STREAM_RBM_H, STREAM_RBM_V, STREAM_DBN_ENC, STREAM_DBN_DEC = range(6)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def bits_to_uniform(x):
    """TF ``random_uniform`` float32 mapping: 23 mantissa bits, [0,1)."""
    x = np.asarray(x, dtype=np.uint32)
    return ((x & np.uint32(0x7FFFFF)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def uniform(seed, stream, row, sub, elem):
    """u for (row, sub, elem) broadcastable index arrays -> float32 array."""
    row, sub, elem = np.broadcast_arrays(np.asarray(row, np.uint32), np.asarray(sub, np.uint32),
                                         np.asarray(elem, np.uint32))
    out = philox4x32_10(elem >> np.uint32(2), row, sub, np.uint32(stream), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = elem & np.uint32(3)
    x = np.where(w == 0, out[0], np.where(w == 1, out[1], np.where(w == 2, out[2], out[3])))
    return bits_to_uniform(x)


def uniform_block(seed, stream, rows, sub, n_elem):
    """[len(rows), n_elem] uniforms for global row ids ``rows`` at sub-counter ``sub``."""
    rows = np.asarray(rows, np.uint32)[:, None]
    elem = np.arange(n_elem, dtype=np.uint32)[None, :]
    return uniform(seed, stream, rows, sub, elem)
