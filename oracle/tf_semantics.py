"""TF 1.13.1 / TFP 0.6.0 op semantics used by the MultINN hot path (oracle, NumPy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function restates one
third-party op the reference calls; the reference call site is cited as
``file:line`` relative to /root/reference/multinn.  These are the "specification
of record" of SURVEY.md section 8(c) items 1-9.
"""
import numpy as np

EPS_SAFE_LOG = 1e-6   # utils/auxiliary.py:9-11
EPS_LOG_LOSS = 1e-7   # tf.losses.log_loss default epsilon (rbm.py:124-128)


def safe_log(x):
    """utils/auxiliary.py:9-11  ``tf.log(1e-6 + tensor)``."""
    return np.log(x.dtype.type(EPS_SAFE_LOG) + x) if isinstance(x, np.ndarray) else np.log(EPS_SAFE_LOG + x)


def sigmoid(x):
    """tf.sigmoid -- 1/(1+exp(-x)), evaluated stably."""
    x = np.asarray(x)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def lstm_block_cell(x, c_prev, h_prev, W, b, forget_bias=0.0):
    """tf.contrib.cudnn_rnn.CudnnCompatibleLSTMCell == LSTMBlockCell(forget_bias=0)
    (models/common/rnn.py:124).

    ``xh=[x,h_prev]; [i,ci,f,o]=xh.W+b; cs=tanh(ci)*sig(i)+cs_prev*sig(f+fb);
    h=tanh(cs)*sig(o)``.  W is ``[in+u, 4u]`` with column blocks ``i|ci|f|o``.
    Returns (h, c, gates_post) where gates_post=(i,g,f,o) after activation.
    """
    u = h_prev.shape[1]
    z = np.concatenate([x, h_prev], axis=1) @ W + b
    i = sigmoid(z[:, 0 * u:1 * u])
    g = np.tanh(z[:, 1 * u:2 * u])
    f = sigmoid(z[:, 2 * u:3 * u] + forget_bias)
    o = sigmoid(z[:, 3 * u:4 * u])
    c = g * i + c_prev * f
    h = np.tanh(c) * o
    return h, c, (i, g, f, o)


def dropout_output(x, keep_prob, u):
    """tf.nn.rnn_cell.DropoutWrapper(output_keep_prob=kp) (models/common/rnn.py:132):
    ``y = x/kp * floor(kp+u)``, u~U[0,1).  ``u`` is supplied (float32 grid).  The add
    ``kp+u`` is a float32 add in TF; it is reproduced in float32 here whatever the
    dtype of x."""
    if keep_prob >= 1.0:
        return x, np.ones_like(x)
    keep = np.floor(np.float32(keep_prob) + u.astype(np.float32)).astype(x.dtype)
    return x / x.dtype.type(keep_prob) * keep, keep


def dense(x, K, b):
    """tf.layers.Dense (models/generators/rnn_nade.py:54-57): ``x.K + b``."""
    return x @ K + b


def sequence_mask(lengths, maxlen):
    return np.arange(maxlen)[None, :] < np.asarray(lengths)[:, None]


def flatten_maybe_padded_sequences(t, lengths=None):
    """utils/sequences.py:6-37: rows (b,t) with t<lengths[b], b-major then t."""
    if lengths is None:
        return t.reshape((-1,) + t.shape[2:])
    m = sequence_mask(lengths, t.shape[1])
    return t[m]


def log_loss(labels, p, eps=EPS_LOG_LOSS):
    """tf.losses.log_loss(reduction=NONE) (rbm.py:124-128, pass_encoder.py:81-85)."""
    return -labels * np.log(p + eps) - (1 - labels) * np.log(1 - p + eps)


def clip_by_global_norm(grads, clip_norm):
    """tf.clip_by_global_norm (utils/training.py:166): ``g*clip/max(gn,clip)``
    written as TF does: ``g * clip * min(1/gn, 1/clip)``."""
    gn = np.sqrt(sum(float(np.sum(np.square(g.astype(np.float64)))) for g in grads))
    scale = clip_norm * min(1.0 / gn, 1.0 / clip_norm) if gn > 0 else 1.0
    return [g * g.dtype.type(scale) for g in grads], gn


def adam_tf_step(theta, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-4):
    """tf.train.AdamOptimizer(lr, epsilon=1e-4) (train.py:64): epsilon OUTSIDE the
    bias correction: ``lr_t=lr*sqrt(1-b2^t)/(1-b1^t); th -= lr_t*m/(sqrt(v)+eps)``.
    ``t`` is the 1-based step count.  Returns (theta, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------- #
# initialisers (rbm.py:36, nade.py:49-50, rnn_nade.py:56)
# --------------------------------------------------------------------------- #
def glorot_uniform(rng, fan_in, fan_out, shape=None, dtype=np.float32):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape or (fan_in, fan_out)).astype(dtype)


def truncated_normal(rng, shape, std, dtype=np.float32):
    """tf.truncated_normal_initializer: resample beyond 2 sigma."""
    x = rng.standard_normal(size=shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(dtype)
