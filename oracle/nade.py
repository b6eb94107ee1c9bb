"""NADE log_prob / backward / sample restatement (oracle, NumPy, dtype-generic).

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/multinn/models/common/nade.py:
``log_prob`` 155-229, ``sample`` 231-308, ``_cond_prob`` 310-329, and
utils/auxiliary.py:9-11 (``safe_log``).  Weight layout here: ``w_enc[D,Hn]``
(reference ``[D,1,Hn]``, nade.py:54-58) and ``w_dec[D,Hn]`` (reference
``[D,Hn,1]``, nade.py:61-66).
"""
import numpy as np

from .tf_semantics import sigmoid, EPS_SAFE_LOG


def log_prob(v, b_enc, b_dec, w_enc, w_dec):
    """nade.py:155-229.  v[N,D] in {0,1}; returns (nll[N], cond_p[N,D])."""
    N, D = v.shape
    dt = b_enc.dtype
    eps = dt.type(EPS_SAFE_LOG)
    a = b_enc.copy()                      # a_0 = b_enc           nade.py:185
    log_p = np.zeros(N, dt)
    cond_p = np.empty((N, D), dt)
    for i in range(D):                    # nade.py:225-226
        h = sigmoid(a)                    # nade.py:326
        l = b_dec[:, i] + h @ w_dec[i]    # nade.py:327
        p = sigmoid(l)                    # nade.py:328
        vi = v[:, i].astype(dt)
        log_p += vi * np.log(eps + p) + (1 - vi) * np.log(eps + (1 - p))   # nade.py:208
        cond_p[:, i] = p
        a = a + vi[:, None] * w_enc[i][None, :]                              # nade.py:219
    return -log_p, cond_p


def log_prob_bwd(v, b_enc, b_dec, w_enc, w_dec, row_weight):
    """Gradient of ``L = sum_n row_weight[n] * nll[n]`` (build-owned derivation,
    SURVEY.md Appendix A.1; autograd-equivalent of nade.py:199-229).

    Returns (d_b_enc[N,Hn], d_b_dec[N,D], d_w_enc[D,Hn], d_w_dec[D,Hn]).
    """
    N, D = v.shape
    dt = b_enc.dtype
    eps = dt.type(EPS_SAFE_LOG)
    # forward, keeping every a_i (oracle only -- the product never stores these)
    a_all = np.empty((D + 1,) + b_enc.shape, dt)
    a_all[0] = b_enc
    for i in range(D):
        a_all[i + 1] = a_all[i] + v[:, i].astype(dt)[:, None] * w_enc[i][None, :]
    d_b_dec = np.empty((N, D), dt)
    d_w_enc = np.zeros_like(w_enc)
    d_w_dec = np.zeros_like(w_dec)
    G = np.zeros_like(b_enc)              # G_{i+1}
    rw = row_weight.astype(dt)
    for i in range(D - 1, -1, -1):
        vi = v[:, i].astype(dt)
        h = sigmoid(a_all[i])
        p = sigmoid(b_dec[:, i] + h @ w_dec[i])
        dnll_dp = -(vi / (eps + p) - (1 - vi) / (eps + (1 - p)))
        dl = rw * dnll_dp * p * (1 - p)
        d_b_dec[:, i] = dl
        d_w_dec[i] = dl @ h
        d_w_enc[i] = vi @ G               # uses G_{i+1}
        G = G + dl[:, None] * w_dec[i][None, :] * h * (1 - h)
    return G, d_b_dec, d_w_enc, d_w_dec


def sample(b_enc, b_dec, w_enc, w_dec, u=None, temperature=1.0):
    """nade.py:231-308.  ``u[N,D]`` supplies the uniforms of TFP Bernoulli
    (``sample = u < sigmoid(logit/T)``); ``temperature=None``
    thresholds at 0.5 (nade.py:278-279).  Returns (samples[N,D], nll[N])."""
    N = b_enc.shape[0]
    D = b_dec.shape[1]
    dt = b_enc.dtype
    eps = dt.type(EPS_SAFE_LOG)
    a = b_enc.copy()
    log_p = np.zeros(N, dt)
    out = np.empty((N, D), dt)
    for i in range(D):
        h = sigmoid(a)
        l = b_dec[:, i] + h @ w_dec[i]
        p = sigmoid(l)
        if temperature is None:
            vi = (p >= 0.5).astype(dt)
        else:
            ps = sigmoid(l / dt.type(temperature))
            vi = (u[:, i].astype(dt) < ps).astype(dt)
        out[:, i] = vi
        log_p += vi * np.log(eps + p) + (1 - vi) * np.log(eps + (1 - p))
        a = a + vi[:, None] * w_enc[i][None, :]
    return out, -log_p
