"""RnnNade / RnnMultiNADE / RnnRBM train step and sampling scan (oracle, NumPy).

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/multinn:
  models/multinn/multinn_joint.py:76-89,132-139   joint input plumbing (A1, A2)
  models/generators/rnn_nade.py:64-124,173-318     LSTM -> Dense -> NADE (A5-A8)
  models/generators/rnn_multinade.py:83-317        shared LSTM + M NADEs (A12)
  models/generators/rnn_rbm.py:71-322              LSTM -> (Wuh,Wuv) -> RBM (A14-A16)
  models/generators/rnn_estimator.py:271-323       generate scan (A11)
  metrics/statistical.py:34                        batch/loss = mean(loss)
  utils/training.py:151-177 + train.py:61-64       clip 5.0 + TF Adam (A9)
"""
import numpy as np

from . import nade, rbm, lstm, philox
from .tf_semantics import (flatten_maybe_padded_sequences, sequence_mask, clip_by_global_norm,
                           adam_tf_step, glorot_uniform, truncated_normal, dense)


# --------------------------------------------------------------------------- #
# plumbing
# --------------------------------------------------------------------------- #
def joint_inputs(x):
    """multinn_joint.py:83-89 + 132-139: [B,T,P,M] -> (inputs[B,T,P*M], targets[B,T,P*M]);
    feature index p*M+m, one all-zero step prepended, inputs=enc[:,:-1], targets=enc[:,1:]."""
    B, T, P, M = x.shape
    flat = x.reshape(B, T, P * M)
    enc = np.concatenate([np.zeros((B, 1, P * M), flat.dtype), flat], axis=1)
    return enc[:, :-1], enc[:, 1:]


def per_track_inputs(x):
    """multi_encoder_nn.py:66-76: pad t=0 then unstack the track axis -> M x [B,T+1,P]."""
    B, T, P, M = x.shape
    enc = np.concatenate([np.zeros((B, 1, P, M), x.dtype), x], axis=1)
    return [enc[..., m] for m in range(M)]


def training_windows(T_total, lengths, piece_size):
    """train.py:165-186 window slicing (A13): for each piece j, the clipped lengths and
    the indices of songs that still have steps.  Returns [(j0, j1, keep_idx, clipped_len)]."""
    out = []
    lengths = np.asarray(lengths)
    for j in range(0, T_total, piece_size):
        seq_len = lengths - j
        keep = np.nonzero(seq_len > 0)[0]
        if keep.size == 0:
            continue
        out.append((j, min(j + piece_size, T_total), keep, np.minimum(seq_len[keep], piece_size)))
    return out


def row_weights(lengths, B, T, dtype):
    """Weights such that sum_n w[n]*nll[n] == mean over valid rows (statistical.py:34)."""
    if lengths is None:
        return np.full(B * T, 1.0 / (B * T), dtype)
    N = int(np.sum(lengths))
    return np.full(N, 1.0 / N, dtype)


# --------------------------------------------------------------------------- #
# parameter construction (reference initialisers)
# --------------------------------------------------------------------------- #
def init_lstm(rng, n_in, units, dtype=np.float32):
    layers = []
    for u in units:
        layers.append((glorot_uniform(rng, n_in + u, 4 * u, dtype=dtype), np.zeros(4 * u, dtype)))
        n_in = u
    return layers


def init_rnn_nade(seed, n_in, D, Hn, units, dtype=np.float32, tracks=1):
    """RnnNade (tracks=1) or RnnMultiNADE (tracks=M, D per track) parameters."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p = dict(lstm=init_lstm(rng, n_in, units, dtype))
    n_out = tracks * (D + Hn)
    p['fc_k'] = glorot_uniform(rng, units[-1], n_out, dtype=dtype)
    p['fc_b'] = np.zeros(n_out, dtype)
    std = 1.0 / np.sqrt(D)
    p['w_enc'] = [truncated_normal(rng, (D, Hn), std, dtype) for _ in range(tracks)]
    p['w_dec'] = [truncated_normal(rng, (D, Hn), std, dtype) for _ in range(tracks)]
    return p


def init_rnn_rbm(seed, n_in, D, Hn, units, dtype=np.float32):
    rng = np.random.Generator(np.random.PCG64(seed))
    p = dict(lstm=init_lstm(rng, n_in, units, dtype))
    p['W'] = glorot_uniform(rng, D, Hn, dtype=dtype)
    p['bh'] = np.zeros((1, Hn), dtype)
    p['bv'] = np.zeros((1, D), dtype)
    p['Wuh'] = glorot_uniform(rng, units[-1], Hn, dtype=dtype)
    p['Wuv'] = glorot_uniform(rng, units[-1], D, dtype=dtype)
    return p


def dropout_uniforms(seed, B, T, units, row0=0):
    """RNG contract: stream 0, row = global batch index, sub = (t<<8)|layer, elem = unit."""
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    out = []
    for l, u in enumerate(units):
        a = np.empty((B, T, u), np.float32)
        for t in range(T):
            a[:, t] = philox.uniform_block(seed, philox.STREAM_DROPOUT, rows, (t << 8) | l, u)
        out.append(a)
    return out


# --------------------------------------------------------------------------- #
# RnnNade / RnnMultiNADE
# --------------------------------------------------------------------------- #
def split_biases(out, Hn, D, tracks=1):
    """rnn_nade.py:245 (b_enc first) / rnn_multinade.py:242-249 (all b_enc blocks, then all b_dec)."""
    b_enc = [out[:, m * Hn:(m + 1) * Hn] for m in range(tracks)]
    off = tracks * Hn
    b_dec = [out[:, off + m * D:off + (m + 1) * D] for m in range(tracks)]
    return b_enc, b_dec


def rnn_nade_forward(inputs, targets, lengths, p, keep_prob=1.0, drop_u=None, tracks=1):
    """rnn_nade.py:279-302 + 91-108 (tracks=1) / rnn_multinade.py:258-293.
    Returns dict(loss, nll[list of N], cond_p[list of [N,D]], cache)."""
    B, T, _ = inputs.shape
    Hn = p['w_enc'][0].shape[1]
    D = p['w_enc'][0].shape[0]
    y, _, cache = lstm.seq_fwd(inputs, p['lstm'], keep_prob, drop_u, lengths, 'decode')
    yf = flatten_maybe_padded_sequences(y, lengths)
    out = dense(yf, p['fc_k'], p['fc_b'])
    b_enc, b_dec = split_biases(out, Hn, D, tracks)
    tf_ = flatten_maybe_padded_sequences(targets, lengths)
    # rnn_multinade.py:97-101: reshape(flat,[-1,E,M]) unstacked on the last axis (track-minor)
    tgt = [tf_] if tracks == 1 else [tf_.reshape(-1, D, tracks)[..., m] for m in range(tracks)]
    nll, cond = [], []
    for m in range(tracks):
        n_, c_ = nade.log_prob(tgt[m], b_enc[m], b_dec[m], p['w_enc'][m], p['w_dec'][m])
        nll.append(n_)
        cond.append(c_)
    loss = np.mean([n_.mean() for n_ in nll])          # rnn_multinade.py:202-203 / statistical.py:34
    return dict(loss=loss, nll=nll, cond_p=cond, cache=cache, yf=yf, b_enc=b_enc, b_dec=b_dec, tgt=tgt,
                lengths=lengths, BT=(B, T))


def rnn_nade_backward(fw, p, tracks=1, n_total=None):
    """Gradients of fw['loss'] wrt every trainable variable (rnn_nade.py:117-120).
    n_total: number of valid rows over ALL data-parallel shards (the loss is a mean over rows,
    statistical.py:34, so a shard's gradient carries weight N_shard/N_total)."""
    B, T = fw['BT']
    lengths = fw['lengths']
    dt = p['fc_k'].dtype
    N = fw['yf'].shape[0]
    rw = np.full(N, 1.0 / ((n_total or N) * tracks), dt)
    d_out = np.zeros((N, p['fc_k'].shape[1]), dt)
    Hn = p['w_enc'][0].shape[1]
    D = p['w_enc'][0].shape[0]
    g = dict(w_enc=[], w_dec=[])
    for m in range(tracks):
        dbe, dbd, dwe, dwd = nade.log_prob_bwd(fw['tgt'][m], fw['b_enc'][m], fw['b_dec'][m],
                                               p['w_enc'][m], p['w_dec'][m], rw)
        d_out[:, m * Hn:(m + 1) * Hn] = dbe
        d_out[:, tracks * Hn + m * D:tracks * Hn + (m + 1) * D] = dbd
        g['w_enc'].append(dwe)
        g['w_dec'].append(dwd)
    g['fc_k'] = fw['yf'].T @ d_out
    g['fc_b'] = d_out.sum(0)
    dyf = d_out @ p['fc_k'].T
    dy = np.zeros((B, T, dyf.shape[1]), dt)
    if lengths is None:
        dy[...] = dyf.reshape(B, T, -1)
    else:
        dy[sequence_mask(lengths, T)] = dyf
    dx, lg = lstm.seq_bwd(dy, fw['cache'])
    g['lstm'] = lg
    g['dx'] = dx
    return g


def flat_params(p):
    """Trainable variables in the order rnn_nade.py:117-120: rnn, nade [w_enc,w_dec], dense."""
    out = []
    for W, b in p['lstm']:
        out += [W, b]
    for m in range(len(p['w_enc'])):
        out += [p['w_enc'][m], p['w_dec'][m]]
    out += [p['fc_k'], p['fc_b']]
    return out


def flat_grads(g):
    out = []
    for W, b in g['lstm']:
        out += [W, b]
    for m in range(len(g['w_enc'])):
        out += [g['w_enc'][m], g['w_dec'][m]]
    out += [g['fc_k'], g['fc_b']]
    return out


def apply_clip_adam(params, grads, opt, lr=0.01, clip=5.0, eps=1e-4):
    """utils/training.py:163-175 + train.py:64.  opt = dict(t, m[], v[]); updates in place."""
    clipped, gn = clip_by_global_norm(grads, clip)
    opt['t'] += 1
    for k, (th, g_) in enumerate(zip(params, clipped)):
        th2, opt['m'][k], opt['v'][k] = adam_tf_step(th, g_, opt['m'][k], opt['v'][k], opt['t'], lr, eps=eps)
        th[...] = th2
    return gn


def new_opt(params):
    return dict(t=0, m=[np.zeros_like(x) for x in params], v=[np.zeros_like(x) for x in params])


def rnn_nade_train_step(inputs, targets, lengths, p, opt, keep_prob, drop_u, lr=0.01, tracks=1):
    fw = rnn_nade_forward(inputs, targets, lengths, p, keep_prob, drop_u, tracks)
    g = rnn_nade_backward(fw, p, tracks)
    gn = apply_clip_adam(flat_params(p), flat_grads(g), opt, lr)
    return fw['loss'], gn, fw, g


# --------------------------------------------------------------------------- #
# sampling scan (rnn_estimator.py:271-323, rnn_nade.py:253-277,304-318)
# --------------------------------------------------------------------------- #
def lstm_single_step(x, state, layers):
    new = []
    inp = x
    for (W, b), (c, h) in zip(layers, state):
        from .tf_semantics import lstm_block_cell
        h2, c2, _ = lstm_block_cell(inp, c, h, W, b)
        new.append((c2, h2))
        inp = h2                                     # is_train False -> keep_prob 1 (rnn.py:120)
    return inp, new


def rnn_nade_generate(intro, num_steps, p, seed, row0=0, tracks=1, temperature=1.0):
    """intro[B,Ti,Din] -> samples[B,num_steps,tracks*D].  Uniforms: stream 1, row = global
    batch index, sub = generated step, elem = m*D+i (track-major inside a step)."""
    B = intro.shape[0]
    Hn = p['w_enc'][0].shape[1]
    D = p['w_enc'][0].shape[0]
    dt = p['fc_k'].dtype
    y, state, _ = lstm.seq_fwd(intro, p['lstm'], 1.0, None, None, 'decode')
    out = dense(y[:, -1], p['fc_k'], p['fc_b'])
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    samples = np.empty((B, num_steps, tracks * D), dt)
    for s in range(num_steps):
        b_enc, b_dec = split_biases(out, Hn, D, tracks)
        u = philox.uniform_block(seed, philox.STREAM_NADE, rows, s, tracks * D)
        per = []
        for m in range(tracks):
            smp, _ = nade.sample(b_enc[m], b_dec[m], p['w_enc'][m], p['w_dec'][m],
                                 u[:, m * D:(m + 1) * D], temperature)
            per.append(smp)
        # rnn_multinade.py:313-314: stack(axis=2) then reshape -> feature index i*M+m (track-minor)
        step = per[0] if tracks == 1 else np.stack(per, axis=2).reshape(B, tracks * D)
        samples[:, s] = step
        h, state = lstm_single_step(step, state, p['lstm'])
        out = dense(h, p['fc_k'], p['fc_b'])
    return samples


# --------------------------------------------------------------------------- #
# RnnRBM (rnn_rbm.py), intended semantics (R1-R4)
# --------------------------------------------------------------------------- #
def gibbs_uniforms(seed, rows, k, Hn, D, sub0=0):
    """stream 2/3, row = global flat row id, sub = sub0 + gibbs iteration."""
    rows = np.asarray(rows, np.uint32)
    u_h = np.stack([philox.uniform_block(seed, philox.STREAM_RBM_H, rows, sub0 + it, Hn) for it in range(k)]) \
        if k else np.zeros((0, len(rows), Hn), np.float32)
    u_v = np.stack([philox.uniform_block(seed, philox.STREAM_RBM_V, rows, sub0 + it, D) for it in range(k)]) \
        if k else np.zeros((0, len(rows), D), np.float32)
    return u_h, u_v


def rnn_rbm_forward(inputs, targets, lengths, p, k, seed, keep_prob=1.0, drop_u=None, bias_mode='conditional',
                    row_ids=None):
    """rnn_rbm.py:71-123 with R2 (lengths forwarded) and R3 (bias_mode).  The chain
    starts from inputs_flat (rnn_rbm.py:112)."""
    B, T, _ = inputs.shape
    y, _, cache = lstm.seq_fwd(inputs, p['lstm'], keep_prob, drop_u, lengths, 'dynamic_rnn')
    yf = flatten_maybe_padded_sequences(y, lengths)
    bh_t = p['bh'] + yf @ p['Wuh']          # rnn_rbm.py:252-257 (internal_bias=True default)
    bv_t = p['bv'] + yf @ p['Wuv']
    v0 = flatten_maybe_padded_sequences(inputs, lengths)
    tgt = flatten_maybe_padded_sequences(targets, lengths)
    N = yf.shape[0]
    rows = np.arange(N) if row_ids is None else row_ids
    u_h, u_v = gibbs_uniforms(seed, rows, k, p['W'].shape[1], p['W'].shape[0])
    p_v, v_s = rbm.gibbs(v0, p['W'], bh_t, bv_t, k, u_h, u_v)
    if bias_mode == 'conditional':
        cost, F = rbm.free_energy_cost(tgt, v_s, p['W'], bh_t, bv_t)
    else:
        cost, F = rbm.free_energy_cost(tgt, v_s, p['W'], p['bh'], p['bv'])
    return dict(loss=cost.mean(), cost=cost, free_energy=F, p_v=p_v, v_sample=v_s,
                recon=rbm.reconstruction_cost(tgt, p_v), cache=cache, yf=yf, bh_t=bh_t, bv_t=bv_t,
                tgt=tgt, lengths=lengths, BT=(B, T), bias_mode=bias_mode)


def rnn_rbm_backward(fw, p):
    B, T = fw['BT']
    dt = p['W'].dtype
    N = fw['yf'].shape[0]
    rw = np.full(N, 1.0 / N, dt)
    g = {}
    if fw['bias_mode'] == 'conditional':
        dW, dbh, dbv = rbm.free_energy_cost_bwd(fw['tgt'], fw['v_sample'], p['W'], fw['bh_t'], fw['bv_t'], rw)
        g['W'], g['bh'], g['bv'] = dW, dbh.sum(0, keepdims=True), dbv.sum(0, keepdims=True)
        g['Wuh'] = fw['yf'].T @ dbh
        g['Wuv'] = fw['yf'].T @ dbv
        dyf = dbh @ p['Wuh'].T + dbv @ p['Wuv'].T
        dy = np.zeros((B, T, dyf.shape[1]), dt)
        if fw['lengths'] is None:
            dy[...] = dyf.reshape(B, T, -1)
        else:
            dy[sequence_mask(fw['lengths'], T)] = dyf
        dx, lg = lstm.seq_bwd(dy, fw['cache'])
        g['lstm'], g['dx'] = lg, dx
    else:
        dW, dbh, dbv = rbm.free_energy_cost_bwd(fw['tgt'], fw['v_sample'], p['W'],
                                                np.broadcast_to(p['bh'], fw['bh_t'].shape),
                                                np.broadcast_to(p['bv'], fw['bv_t'].shape), rw)
        g['W'], g['bh'], g['bv'] = dW, dbh.sum(0, keepdims=True), dbv.sum(0, keepdims=True)
    return g


# --------------------------------------------------------------------------- #
# Feedback-RNN sampling scan (multinn_feedback.py:120-218, multinn_feedback_rnn.py:41-79), A19
# --------------------------------------------------------------------------- #
def feedback_rnn_generate(x, num_steps, gen_params, fb_layers, seeds, row0=0):
    """x [B,Ti,P,M] -> samples [B,num_steps,P,M].  gen_params[i]: RnnNade params with n_in = P+F;
    fb_layers: feedback LSTM layers over P*M inputs; seeds[i]: RNG seed of generator i
    (uniforms: stream 1, row = global batch index, sub = step, elem = visible)."""
    B, Ti, P, M = x.shape
    dt = gen_params[0]['fc_k'].dtype
    enc = np.concatenate([np.zeros((B, 1, P, M), dt), x.astype(dt)], axis=1)
    stack = enc.reshape(B, Ti + 1, P * M)
    x_fb, fb_state, _ = lstm.seq_fwd(stack, fb_layers)
    Hn = gen_params[0]['w_enc'][0].shape[1]
    states, outs = [], []
    for i, p in enumerate(gen_params):
        y, st, _ = lstm.seq_fwd(np.concatenate([enc[..., i], x_fb], -1), p['lstm'])
        states.append(st)
        outs.append(dense(y[:, -1], p['fc_k'], p['fc_b']))
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    samples = np.empty((B, num_steps, P, M), dt)
    probs = np.empty((B, num_steps, P, M), dt)
    for s in range(num_steps):
        cur = []
        for i, p in enumerate(gen_params):
            b_enc, b_dec = split_biases(outs[i], Hn, P, 1)
            u = philox.uniform_block(seeds[i], philox.STREAM_NADE, rows, s, P)
            smp, _ = nade.sample(b_enc[0], b_dec[0], p['w_enc'][0], p['w_dec'][0], u, 1.0)
            _, cp = nade.log_prob(smp, b_enc[0], b_dec[0], p['w_enc'][0], p['w_dec'][0])
            probs[:, s, :, i] = cp
            cur.append(smp)
        st = np.stack(cur, -1)
        samples[:, s] = st
        fb, fb_state = lstm_single_step(st.reshape(B, P * M), fb_state, fb_layers)
        for i, p in enumerate(gen_params):
            h, states[i] = lstm_single_step(np.concatenate([cur[i], fb], 1), states[i], p['lstm'])
            outs[i] = dense(h, p['fc_k'], p['fc_b'])
    return samples, probs


def feedback_rnn_teacher_forced(x, samples, gen_params, fb_layers, seeds, row0=0):
    """Replay the scan on GIVEN samples: returns (probs, uniforms) so a device scan can be verified draw by draw."""
    B, Ti, P, M = x.shape
    dt = gen_params[0]['fc_k'].dtype
    enc = np.concatenate([np.zeros((B, 1, P, M), dt), x.astype(dt)], axis=1)
    x_fb, fb_state, _ = lstm.seq_fwd(enc.reshape(B, Ti + 1, P * M), fb_layers)
    Hn = gen_params[0]['w_enc'][0].shape[1]
    states, outs = [], []
    for i, p in enumerate(gen_params):
        y, st, _ = lstm.seq_fwd(np.concatenate([enc[..., i], x_fb], -1), p['lstm'])
        states.append(st)
        outs.append(dense(y[:, -1], p['fc_k'], p['fc_b']))
    rows = np.arange(row0, row0 + B, dtype=np.uint32)
    S = samples.shape[1]
    probs, us = np.empty((B, S, P, M), dt), np.empty((B, S, P, M), np.float32)
    for s in range(S):
        cur = [samples[:, s, :, i].astype(dt) for i in range(M)]
        for i, p in enumerate(gen_params):
            b_enc, b_dec = split_biases(outs[i], Hn, P, 1)
            _, probs[:, s, :, i] = nade.log_prob(cur[i], b_enc[0], b_dec[0], p['w_enc'][0], p['w_dec'][0])
            us[:, s, :, i] = philox.uniform_block(seeds[i], philox.STREAM_NADE, rows, s, P)
        fb, fb_state = lstm_single_step(np.stack(cur, -1).reshape(B, P * M), fb_state, fb_layers)
        for i, p in enumerate(gen_params):
            h, states[i] = lstm_single_step(np.concatenate([cur[i], fb], 1), states[i], p['lstm'])
            outs[i] = dense(h, p['fc_k'], p['fc_b'])
    return probs, us
