"""Multi-layer LSTM over a sequence, forward + BPTT (oracle, NumPy, dtype-generic).

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/multinn/models/common/rnn.py:
104-145 (``MultiRNNCell`` of ``DropoutWrapper(CudnnCompatibleLSTMCell)``), with the
two loop flavours the reference uses: ``dynamic_decode(impute_finished=False)``
(models/generators/rnn_nade.py:204-218) and ``tf.nn.dynamic_rnn(sequence_length)``
(models/generators/rnn_rbm.py:217-223).  TF op semantics: oracle/tf_semantics.py.
"""
import numpy as np

from .tf_semantics import lstm_block_cell, dropout_output


def zero_state(layers, B, dtype):
    """rnn.py:155-176 (learn_zero_state=False)."""
    return [(np.zeros((B, W.shape[1] // 4), dtype), np.zeros((B, W.shape[1] // 4), dtype)) for W, _ in layers]


def seq_fwd(x, layers, keep_prob=1.0, drop_u=None, lengths=None, flavour='decode', init_state=None):
    """x[B,T,in]; layers = [(W[in+u,4u], b[4u]), ...]; drop_u[l][B,T,u] uniforms.

    flavour 'decode': every (b,t) is stepped (impute_finished=False).
    flavour 'dynamic_rnn': for t>=lengths[b] output is zero and state is copied through.
    Returns (y[B,T,u_last], final_state, cache).
    """
    B, T, _ = x.shape
    dt = x.dtype
    state = init_state or zero_state(layers, B, dt)
    state = [(c.copy(), h.copy()) for c, h in state]
    L = len(layers)
    cache = dict(x=x, keep_prob=keep_prob, layers=layers,
                 gates=[[None] * T for _ in range(L)], c=[[None] * T for _ in range(L)],
                 h=[[None] * T for _ in range(L)], inp=[[None] * T for _ in range(L)],
                 keep=[[None] * T for _ in range(L)], c0=[s[0] for s in state], h0=[s[1] for s in state],
                 live=None)
    y = np.zeros((B, T, layers[-1][0].shape[1] // 4), dt)
    live_all = np.ones((B, T), bool)
    if flavour == 'dynamic_rnn' and lengths is not None:
        live_all = np.arange(T)[None, :] < np.asarray(lengths)[:, None]
    cache['live'] = live_all
    for t in range(T):
        inp = x[:, t]
        live = live_all[:, t][:, None]
        for l, (W, b) in enumerate(layers):
            c_prev, h_prev = state[l]
            h, c, gates = lstm_block_cell(inp, c_prev, h_prev, W, b)
            u = None if drop_u is None else drop_u[l][:, t]
            out, keep = dropout_output(h, keep_prob, u) if keep_prob < 1.0 else (h, np.ones_like(h))
            cache['gates'][l][t], cache['c'][l][t], cache['h'][l][t] = gates, c, h
            cache['inp'][l][t], cache['keep'][l][t] = inp, keep
            # dynamic_rnn: copy state through / zero output past the length
            state[l] = (np.where(live, c, c_prev), np.where(live, h, h_prev))
            inp = np.where(live, out, 0)
        y[:, t] = inp
    return y, state, cache


def seq_bwd(dy, cache):
    """BPTT for seq_fwd (flavour 'decode' or fully-live 'dynamic_rnn' rows only).
    dy[B,T,u_last].  Returns (dx[B,T,in], [(dW,db), ...])."""
    layers = cache['layers']
    kp = cache['keep_prob']
    x = cache['x']
    B, T, _ = x.shape
    dt = x.dtype
    L = len(layers)
    grads = [(np.zeros_like(W), np.zeros_like(b)) for W, b in layers]
    dx = np.zeros_like(x)
    dh_next = [np.zeros_like(cache['h0'][l]) for l in range(L)]
    dc_next = [np.zeros_like(cache['c0'][l]) for l in range(L)]
    live_all = cache['live']
    for t in range(T - 1, -1, -1):
        dout = dy[:, t]
        live = live_all[:, t][:, None]
        for l in range(L - 1, -1, -1):
            W, b = layers[l]
            u = W.shape[1] // 4
            i, g, f, o = cache['gates'][l][t]
            c = cache['c'][l][t]
            c_prev = cache['c'][l][t - 1] if t > 0 else cache['c0'][l]
            h_prev = cache['h'][l][t - 1] if t > 0 else cache['h0'][l]
            dout = np.where(live, dout, 0)
            dh = dout / dt.type(kp) * cache['keep'][l][t] if kp < 1.0 else dout
            dh = dh + np.where(live, dh_next[l], 0)
            tc = np.tanh(c)
            do = dh * tc
            dc = dh * o * (1 - tc * tc) + np.where(live, dc_next[l], 0)
            dz = np.concatenate([dc * g * i * (1 - i), dc * i * (1 - g * g),
                                 dc * c_prev * f * (1 - f), do * o * (1 - o)], axis=1)
            xh = np.concatenate([cache['inp'][l][t], h_prev], axis=1)
            grads[l][0][...] += xh.T @ dz
            grads[l][1][...] += dz.sum(0)
            dxh = dz @ W.T
            nin = W.shape[0] - u
            # rows past their length (dynamic_rnn) just pass the carried grads through
            dh_next[l] = np.where(live, dxh[:, nin:], dh_next[l])
            dc_next[l] = np.where(live, dc * f, dc_next[l])
            dout = dxh[:, :nin]
        dx[:, t] = dout
    return dx, grads
