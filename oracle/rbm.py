"""RBM Gibbs chain / free energy / CD-k restatement (oracle, NumPy, dtype-generic).

TEST INFRASTRUCTURE ONLY.  Follows /root/reference/multinn/models/common/rbm.py:
``forward`` 148-167, ``reconstruct`` 169-190, ``sample`` 192-231,
``free_energy_cost`` 233-263, ``visible_bias_init_ops`` 286-297, ``_cd_update``
299-335, ``_cond_prob_h/_v`` 337-373, ``_sample`` 375-387, and the metric in
``build_metrics`` 96-146.  Reference defects R1-R4 are
resolved as recorded there: per-row free energy, k = rbm.k, bias_mode switch.

Bernoulli draws follow TFP 0.6: ``sample = float(u < p)`` with supplied uniforms
``u_h[k,N,Hn]`` / ``u_v[k,N,D]``.
"""
import numpy as np

from .tf_semantics import sigmoid, log_loss, EPS_SAFE_LOG


def softplus(x):
    """log(1+exp(x)) (rbm.py:258), evaluated without overflow."""
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def cond_prob_h(v, W, bh):
    return sigmoid(v @ W + bh)            # rbm.py:351-352


def cond_prob_v(h, W, bv):
    return sigmoid(h @ W.T + bv)          # rbm.py:370-371


def gibbs(v0, W, bh, bv, k, u_h, u_v):
    """rbm.py:192-231.  Returns (p_v_k[N,D], v_k[N,D]).  k=0 returns (v0, v0)."""
    dt = W.dtype
    p_v, v = v0.astype(dt), v0.astype(dt)
    for it in range(k):
        p_h = cond_prob_h(v, W, bh)
        h = (u_h[it].astype(dt) < p_h).astype(dt)
        p_v = cond_prob_v(h, W, bv)
        v = (u_v[it].astype(dt) < p_v).astype(dt)
    return p_v, v


def free_energy(v, W, bh, bv):
    """Per-row F[n] = -sum_j log(1+exp((vW)_j+bh_j)) - v.bv  (rbm.py:256-258, R4)."""
    bvb = np.broadcast_to(bv, v.shape)
    return -softplus(v @ W + bh).sum(1) - (v * bvb).sum(1)


def free_energy_cost(v, v_sample, W, bh, bv):
    """rbm.py:233-263: cost[n] = F(v)[n] - F(v_sample)[n]; returns (cost, F(v))."""
    Fv = free_energy(v, W, bh, bv)
    return Fv - free_energy(v_sample, W, bh, bv), Fv


def free_energy_cost_as_written(v, v_sample, W, bh1, bv1):
    """The as-written ``[N] - [N,1]`` broadcast of rbm.py:258 with the INTERNAL
    biases bh1[1,Hn], bv1[1,D] (rbm.py:119): returns the [N,N] matrices (cost, F)."""
    def F(vv):
        return -softplus(vv @ W + bh1).sum(1)[None, :] - (vv @ bv1.T)
    Fv = F(v)
    return Fv - F(v_sample), Fv


def free_energy_cost_bwd(v, v_sample, W, bh, bv, row_weight):
    """Gradient of ``sum_n rw[n]*(F(v)-F(v_s))[n]`` with v_s constant (stop_gradient,
    rbm.py:229).  Returns (dW[D,Hn], dbh[N,Hn], dbv[N,D])."""
    dt = W.dtype
    rw = row_weight.astype(dt)[:, None]
    sv = cond_prob_h(v, W, bh)
    ss = cond_prob_h(v_sample, W, bh)
    dbh = rw * (ss - sv)
    dbv = rw * (v_sample - v)
    dW = (rw * v_sample).T @ ss - (rw * v).T @ sv
    return dW, dbh, dbv


def reconstruction_cost(targets, p_v):
    """rbm.py:122-129: sum_d log_loss(target, cond_prob), eps 1e-7."""
    return log_loss(targets, p_v).sum(1)


def visible_bias_init(v):
    """rbm.py:286-297."""
    p = v.mean(0)
    return np.log(v.dtype.type(EPS_SAFE_LOG) + p / (1 - p))[None, :]


def cd_update(v, W, bh1, bv1, k, lr, u_h, u_v, u_h0, u_hk):
    """rbm.py:299-335.  bh1[1,Hn], bv1[1,D] internal biases.  Extra uniforms:
    u_h0[N,Hn] for ``forward(v)`` (rbm.py:316) and u_hk[N,Hn] for
    ``forward(v_sample)`` (rbm.py:319; the sample is drawn but unused).
    Returns (dW, dbv, dbh) -- the quantities that are ``assign_add``-ed."""
    dt = W.dtype
    N = v.shape[0]
    p_v_s, v_s = gibbs(v, W, bh1, bv1, k, u_h, u_v)
    h = (u_h0.astype(dt) < cond_prob_h(v, W, bh1)).astype(dt)
    p_h_s = cond_prob_h(v_s, W, bh1)
    lrn = dt.type(lr) / dt.type(N)
    dW = lrn * (v.T @ h - p_v_s.T @ p_h_s)
    dbv = lrn * (v - p_v_s).sum(0, keepdims=True)
    dbh = lrn * (h - p_h_s).sum(0, keepdims=True)
    return dW, dbv, dbh
