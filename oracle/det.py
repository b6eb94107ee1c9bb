"""ctypes front end of oracle/det_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libdet_ref.so")
_lib = None


def build():
    src = os.path.join(HERE, "det_ref.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "_build/libdet_ref.so"])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sigmoid(x):
    x = f32(x)
    out = np.empty_like(x)
    lib().det_sigmoid_array(_p(x), _p(out), C.c_long(x.size))
    return out


def nade_sample(bias, w_enc, w_dec, tracks, m, D, Hn, temperature, u):
    bias, w_enc, w_dec, u = f32(bias), f32(w_enc), f32(w_dec), f32(u)
    N = bias.shape[0]
    s = np.empty((N, D), np.uint8)
    p = np.empty((N, D), np.float32)
    lib().nade_sample_det(N, D, Hn, tracks, m, _p(bias), bias.shape[1], _p(w_enc), _p(w_dec),
                          C.c_float(-1.0 if temperature is None else temperature), _p(u), _p(s), _p(p))
    return s, p


def rbm_hidden(v, W, bh):
    v, W, bh = f32(v), f32(W), f32(bh)
    N, D = v.shape
    Hn = W.shape[1]
    out = np.empty((N, Hn), np.float32)
    lib().rbm_hidden_det(N, D, Hn, _p(v), _p(W), _p(bh), 0 if bh.shape[0] == 1 else bh.shape[1], _p(out))
    return out


def rbm_visible(h, W, bv):
    h, W, bv = f32(h), f32(W), f32(bv)
    N, Hn = h.shape
    D = W.shape[0]
    out = np.empty((N, D), np.float32)
    lib().rbm_visible_det(N, D, Hn, _p(h), _p(W), _p(bv), 0 if bv.shape[0] == 1 else bv.shape[1], _p(out))
    return out


def rbm_gibbs(v0, W, bh, bv, k, u_h, u_v):
    v0 = np.ascontiguousarray(v0, np.uint8)
    W, bh, bv, u_h, u_v = f32(W), f32(bh), f32(bv), f32(u_h), f32(u_v)
    N, D = v0.shape
    Hn = W.shape[1]
    p_v = np.empty((N, D), np.float32)
    v = np.empty((N, D), np.uint8)
    vb, hb, pb = np.empty((N, D), np.float32), np.empty((N, Hn), np.float32), np.empty((N, Hn), np.float32)
    lib().rbm_gibbs_det(N, D, Hn, k, _p(v0), _p(W), _p(bh), 0 if bh.shape[0] == 1 else bh.shape[1], _p(bv),
                        0 if bv.shape[0] == 1 else bv.shape[1], _p(u_h), _p(u_v), _p(p_v), _p(v), _p(vb), _p(hb), _p(pb))
    return p_v, v


# ---------------------------------------------------------------------------------------------------------------------------------
# Whole sampling scans in the deterministic float32 arithmetic (rnn_estimator.py:271-323; multinn_feedback.py:120-218): every LSTM step,
# Dense and NADE conditional through det_ref.c, uniforms from oracle/philox.py -- the device's generate() must reproduce EVERY cell.
def lstm_step(x, state, layers):
    """One step of the LSTM stack (rnn.py:104-145, is_train False: no dropout).  x [B, n_in] (any dtype, used as float32); state = list of
    (c, h) float32 or None; layers = [(W [(n_in + u), 4u], b [4u])].  Returns (h_top, new_state)."""
    inp = f32(x)
    new = []
    for l, (W, b) in enumerate(layers):
        W, b = f32(W), f32(b)
        u = b.shape[0] // 4
        B, n_in = inp.shape
        assert W.shape == (n_in + u, 4 * u), (W.shape, n_in, u)
        c_prev, h_prev = (None, None) if state is None else (f32(state[l][0]), f32(state[l][1]))
        c, h = np.empty((B, u), np.float32), np.empty((B, u), np.float32)
        lib().lstm_step_det(B, n_in, u, _p(inp), n_in, _p(h_prev) if h_prev is not None else None,
                            _p(c_prev) if c_prev is not None else None, _p(W), _p(b), _p(c), _p(h))
        new.append((c, h))
        inp = h
    return inp, new


def dense(x, W, b=None):
    x, W = f32(x), f32(W)
    B, K = x.shape
    N = W.shape[1]
    out = np.empty((B, N), np.float32)
    lib().dense_det(B, K, N, _p(x), K, _p(W), _p(f32(b)) if b is not None else None, _p(out), N)
    return out


def rnn_nade_generate(intro, num_steps, p, seed, row0=0, tracks=1, temperature=1.0):
    """oracle.generators.rnn_nade_generate in the deterministic float32 arithmetic.  intro [B, Ti, Din] -> samples u8 [B, num_steps, tracks * D]."""
    from . import philox
    B, Ti, _ = intro.shape
    D, Hn = p['w_enc'][0].shape
    state, h = None, None
    for t in range(Ti):
        h, state = lstm_step(intro[:, t], state, p['lstm'])
    out = dense(h, p['fc_k'], p['fc_b'])
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    samples = np.empty((B, num_steps, tracks * D), np.uint8)
    for s in range(num_steps):
        u = philox.uniform_block(seed, philox.STREAM_NADE, rows, s, tracks * D)
        per = [nade_sample(out, p['w_enc'][m], p['w_dec'][m], tracks, m, D, Hn, temperature, u[:, m * D:(m + 1) * D])[0] for m in range(tracks)]
        step = per[0] if tracks == 1 else np.stack(per, axis=2).reshape(B, tracks * D)      # rnn_multinade.py:313-314: feature i * M + m
        samples[:, s] = step
        h, state = lstm_step(step, state, p['lstm'])
        out = dense(h, p['fc_k'], p['fc_b'])
    return samples


def feedback_rnn_generate(x, num_steps, gen_params, fb_layers, seeds, row0=0):
    """oracle.generators.feedback_rnn_generate in the deterministic float32 arithmetic.  x u8 [B, Ti, P, M] -> samples u8 [B, num_steps, P, M]."""
    from . import philox
    B, Ti, P, M = x.shape
    enc = np.concatenate([np.zeros((B, 1, P, M), np.float32), x.astype(np.float32)], axis=1)        # multi_encoder_nn.py:73-76
    stack = enc.reshape(B, Ti + 1, P * M)
    fb_state, states = None, [None] * M
    hs = [None] * M
    for t in range(Ti + 1):
        fb, fb_state = lstm_step(stack[:, t], fb_state, fb_layers)
        for i, p in enumerate(gen_params):
            hs[i], states[i] = lstm_step(np.concatenate([enc[:, t, :, i], fb], 1), states[i], p['lstm'])
    outs = [dense(hs[i], p['fc_k'], p['fc_b']) for i, p in enumerate(gen_params)]
    Hn = gen_params[0]['w_enc'][0].shape[1]
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    samples = np.empty((B, num_steps, P, M), np.uint8)
    for s in range(num_steps):
        cur = []
        for i, p in enumerate(gen_params):
            u = philox.uniform_block(seeds[i], philox.STREAM_NADE, rows, s, P)
            cur.append(nade_sample(outs[i], p['w_enc'][0], p['w_dec'][0], 1, 0, P, Hn, 1.0, u)[0])
        st = np.stack(cur, -1)
        samples[:, s] = st
        fb, fb_state = lstm_step(st.reshape(B, P * M), fb_state, fb_layers)
        for i, p in enumerate(gen_params):
            hs[i], states[i] = lstm_step(np.concatenate([cur[i].astype(np.float32), fb], 1), states[i], p['lstm'])
            outs[i] = dense(hs[i], p['fc_k'], p['fc_b'])
    return samples


def rnn_rbm_generate(intro, num_steps, p, k, seed, row0=0, internal_bias=True):
    """RnnRBM.generate (rnn_estimator.py:271-298 with rnn_rbm.py:240-297; R1: k = rbm.k) in the deterministic float32 arithmetic: intro pass,
    then per generated step a k-step Gibbs chain from the previous step's visibles (uniforms: streams 2 / 3, row = global batch index, sub =
    step * k + iteration) -> LSTM step on the sample -> bh_t = bh + h . Wuh, bv_t = bv + h . Wuv.  intro u8 [B, Ti, D] -> samples u8 [B, num_steps, D]."""
    from . import generators as G
    B, Ti, D = intro.shape
    Hn = p['W'].shape[1]
    state, h = None, None
    for t in range(Ti):
        h, state = lstm_step(intro[:, t], state, p['lstm'])
    bh0 = f32(p['bh']).reshape(-1) if internal_bias else None
    bv0 = f32(p['bv']).reshape(-1) if internal_bias else None
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    prev = np.ascontiguousarray(intro[:, -1], np.uint8)
    out = np.empty((B, num_steps, D), np.uint8)
    for s in range(num_steps):
        bh_t, bv_t = dense(h, p['Wuh'], bh0), dense(h, p['Wuv'], bv0)
        u_h, u_v = G.gibbs_uniforms(seed, rows, k, Hn, D, sub0=s * max(k, 1))
        _, v = rbm_gibbs(prev, p['W'], bh_t, bv_t, k, u_h, u_v)
        out[:, s] = v
        h, state = lstm_step(v, state, p['lstm'])
        prev = v
    return out
