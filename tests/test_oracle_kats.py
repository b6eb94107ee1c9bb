"""Known-answer tests that pin the oracle.

The reference has no tests/fixtures for this path ("parity unpinned"), so the oracle is
pinned by closed forms, enumeration, finite differences and agreement between two
independent restatements (NumPy float64 loops vs torch float32 autograd).
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import nade, rbm, lstm, generators as G, philox, torch_ref as TR
from oracle import tf_semantics as S

R = np.random.Generator(np.random.PCG64(7))


def rand_nade(N, D, Hn, dt=np.float64, rho=0.3):
    v = (R.random((N, D)) < rho).astype(dt)
    return (v, R.standard_normal((N, Hn)).astype(dt) * .5, R.standard_normal((N, D)).astype(dt) * .5,
            R.standard_normal((D, Hn)).astype(dt) * .3, R.standard_normal((D, Hn)).astype(dt) * .3)


# ------------------------------ Philox ------------------------------------- #
def test_philox_random123_kat():
    def h(*a):
        return [int(x) for x in philox.philox4x32_10(*a)]
    assert h(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert h(f, f, f, f, f, f) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert h(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_mapping_grid():
    assert philox.bits_to_uniform(np.uint32(0)) == 0.0
    assert philox.bits_to_uniform(np.uint32(0xffffffff)) == np.float32(1 - 2.0 ** -23)
    u = philox.uniform_block(23, 1, np.arange(64), 5, 440)
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1
    assert abs(u.mean() - 0.5) < 0.01


# ------------------------------ NADE --------------------------------------- #
def test_K1_nade_zero_weights():
    D, Hn, N = 440, 16, 3
    v = (R.random((N, D)) < .5).astype(np.float64)
    z = np.zeros
    nll, p = nade.log_prob(v, z((N, Hn)), z((N, D)), z((D, Hn)), z((D, Hn)))
    assert np.allclose(p, 0.5)
    assert np.allclose(nll, -D * np.log(0.500001), rtol=1e-12)
    assert abs(nll[0] - 304.98388) < 1e-4


def test_K2_nade_closed_form_D2():
    Hn = 3
    be, bd = R.standard_normal((1, Hn)), R.standard_normal((1, 2))
    we, wd = R.standard_normal((2, Hn)), R.standard_normal((2, Hn))
    sg = lambda x: 1 / (1 + np.exp(-x))
    for v0, v1 in itertools.product([0., 1.], repeat=2):
        p0 = sg(bd[0, 0] + sg(be[0]) @ wd[0])
        p1 = sg(bd[0, 1] + sg(be[0] + v0 * we[0]) @ wd[1])
        ref = -(v0 * np.log(1e-6 + p0) + (1 - v0) * np.log(1e-6 + 1 - p0)
                + v1 * np.log(1e-6 + p1) + (1 - v1) * np.log(1e-6 + 1 - p1))
        nll, p = nade.log_prob(np.array([[v0, v1]]), be, bd, we, wd)
        assert np.allclose(nll, ref, rtol=1e-13) and np.allclose(p, [[p0, p1]], rtol=1e-13)


def test_K3_nade_normalisation(monkeypatch):
    D, Hn = 10, 5
    monkeypatch.setattr(nade, 'EPS_SAFE_LOG', 0.0)
    _, be, bd, we, wd = rand_nade(1, D, Hn)
    allv = np.array(list(itertools.product([0., 1.], repeat=D)))
    N = allv.shape[0]
    nll, _ = nade.log_prob(allv, np.repeat(be, N, 0), np.repeat(bd, N, 0), we, wd)
    assert abs(np.exp(-nll).sum() - 1) < 1e-10


def test_K4_nade_order_sensitive():
    v, be, bd, we, wd = rand_nade(4, 12, 6)
    perm = R.permutation(12)
    a, _ = nade.log_prob(v, be, bd, we, wd)
    b, _ = nade.log_prob(v[:, perm], be, bd[:, perm], we[perm], wd[perm])
    # the autoregressive factorisation depends on the visible order (nade.py:225-226)
    assert not np.allclose(a, b)


def test_K5_nade_sample_extremes_and_consistency():
    _, be, bd, we, wd = rand_nade(5, 20, 8)
    ones, _ = nade.sample(be, bd, we, wd, np.zeros((5, 20), np.float32))
    assert (ones == 1).all()
    zeros, _ = nade.sample(be, bd, we, wd, np.full((5, 20), 1 - 2.0 ** -23, np.float32))
    assert (zeros == 0).all()
    u = philox.uniform_block(1, 1, np.arange(5), 0, 20)
    s, nll = nade.sample(be, bd, we, wd, u)
    nll2, _ = nade.log_prob(s, be, bd, we, wd)
    assert np.allclose(nll, nll2, rtol=1e-13)
    thr, _ = nade.sample(be, bd, we, wd, None, temperature=None)
    _, p = nade.log_prob(thr, be, bd, we, wd)
    assert ((p >= .5) == (thr == 1)).all()


def test_K6_nade_finite_difference_grads():
    v, be, bd, we, wd = rand_nade(3, 7, 4)
    rw = R.random(3)
    loss = lambda *a: float((rw * nade.log_prob(v, *a)[0]).sum())
    g = nade.log_prob_bwd(v, be, bd, we, wd, rw)
    args = [be, bd, we, wd]
    for k, arr in enumerate(args):
        for idx in list(np.ndindex(arr.shape))[::3]:
            e = 1e-6
            ap = [x.copy() for x in args]
            am = [x.copy() for x in args]
            ap[k][idx] += e
            am[k][idx] -= e
            fd = (loss(*ap) - loss(*am)) / (2 * e)
            assert abs(fd - g[k][idx]) < 1e-6 * max(1, abs(fd)), (k, idx, fd, g[k][idx])


# ------------------------------ RBM ---------------------------------------- #
def test_K7_rbm_zero_weights():
    D, Hn, N = 9, 6, 4
    v = (R.random((N, D)) < .5).astype(np.float64)
    bv = R.standard_normal((N, D))
    F = rbm.free_energy(v, np.zeros((D, Hn)), np.zeros((N, Hn)), bv)
    assert np.allclose(F, -Hn * np.log(2) - (v * bv).sum(1))


def test_K8_rbm_partition_function():
    D, Hn = 8, 5
    W, bh, bv = R.standard_normal((D, Hn)) * .5, R.standard_normal((1, Hn)), R.standard_normal((1, D))
    allv = np.array(list(itertools.product([0., 1.], repeat=D)))
    allh = np.array(list(itertools.product([0., 1.], repeat=Hn)))
    Zf = np.exp(-rbm.free_energy(allv, W, bh, bv)).sum()
    E = -(allv @ W @ allh.T) - (allv @ bv.T) - (allh @ bh.T).T
    assert np.allclose(Zf, np.exp(-E).sum(), rtol=1e-12)


def test_K9_rbm_gibbs_hand_chain():
    W = np.array([[2., -1.], [0.5, 1.]])
    bh, bv = np.array([[0.1, -0.2]]), np.array([[0., 0.3]])
    sg = lambda x: 1 / (1 + np.exp(-x))
    v = np.array([[1., 0.]])
    u_h = np.array([[[.5, .5]], [[.9, .1]]], np.float32)
    u_v = np.array([[[.6, .6]], [[.2, .99]]], np.float32)
    # step 1
    ph = sg(v @ W + bh); h = (u_h[0] < ph).astype(float)
    pv = sg(h @ W.T + bv); v1 = (u_v[0] < pv).astype(float)
    ph = sg(v1 @ W + bh); h = (u_h[1] < ph).astype(float)
    pv2 = sg(h @ W.T + bv); v2 = (u_v[1] < pv2).astype(float)
    p, s = rbm.gibbs(v, W, bh, bv, 2, u_h, u_v)
    assert np.array_equal(s, v2) and np.allclose(p, pv2)
    p0, s0 = rbm.gibbs(v, W, bh, bv, 0, u_h, u_v)
    assert np.array_equal(s0, v)


def test_K10_rbm_as_written_mean_equals_per_row_mean():
    D, Hn, N = 7, 4, 6
    W, bh, bv = R.standard_normal((D, Hn)), R.standard_normal((1, Hn)), R.standard_normal((1, D))
    v = (R.random((N, D)) < .5).astype(float)
    vs = (R.random((N, D)) < .5).astype(float)
    cost_nn, F_nn = rbm.free_energy_cost_as_written(v, vs, W, bh, bv)
    cost, F = rbm.free_energy_cost(v, vs, W, bh, bv)
    assert cost_nn.shape == (N, N)
    assert np.allclose(cost_nn.mean(), cost.mean()) and np.allclose(F_nn.mean(), F.mean())


def test_K11_rbm_cd_update_closed_form_N1():
    D, Hn = 5, 3
    W, bh, bv = R.standard_normal((D, Hn)), R.standard_normal((1, Hn)), R.standard_normal((1, D))
    v = (R.random((1, D)) < .5).astype(float)
    k = 2
    u_h, u_v = R.random((k, 1, Hn)).astype(np.float32), R.random((k, 1, D)).astype(np.float32)
    u0, uk = R.random((1, Hn)).astype(np.float32), R.random((1, Hn)).astype(np.float32)
    dW, dbv, dbh = rbm.cd_update(v, W, bh, bv, k, 0.1, u_h, u_v, u0, uk)
    pv, vs = rbm.gibbs(v, W, bh, bv, k, u_h, u_v)
    h = (u0 < rbm.cond_prob_h(v, W, bh)).astype(float)
    phs = rbm.cond_prob_h(vs, W, bh)
    assert np.allclose(dW, 0.1 * (np.outer(v, h) - np.outer(pv, phs)))
    assert np.allclose(dbv, 0.1 * (v - pv)) and np.allclose(dbh, 0.1 * (h - phs))


def test_rbm_free_energy_grad_fd():
    D, Hn, N = 5, 4, 3
    W = R.standard_normal((D, Hn)); bh = R.standard_normal((N, Hn)); bv = R.standard_normal((N, D))
    v = (R.random((N, D)) < .5).astype(float); vs = (R.random((N, D)) < .5).astype(float)
    rw = R.random(N)
    f = lambda W_, bh_, bv_: float((rw * rbm.free_energy_cost(v, vs, W_, bh_, bv_)[0]).sum())
    g = rbm.free_energy_cost_bwd(v, vs, W, bh, bv, rw)
    for k, arr in enumerate([W, bh, bv]):
        for idx in np.ndindex(arr.shape):
            a = [W.copy(), bh.copy(), bv.copy()]; b = [W.copy(), bh.copy(), bv.copy()]
            a[k][idx] += 1e-6; b[k][idx] -= 1e-6
            fd = (f(*a) - f(*b)) / 2e-6
            assert abs(fd - g[k][idx]) < 1e-6


# ------------------------------ LSTM --------------------------------------- #
def test_K12_lstm_zero_weights():
    x = R.standard_normal((2, 5, 3))
    layers = [(np.zeros((3 + 4, 16)), np.zeros(16)), (np.zeros((4 + 2, 8)), np.zeros(8))]
    y, st, _ = lstm.seq_fwd(x, layers)
    assert (y == 0).all() and all((c == 0).all() and (h == 0).all() for c, h in st)


def test_K13_lstm_matches_torch_nn_lstm():
    B, T, nin, u = 3, 6, 5, 4
    x = R.standard_normal((B, T, nin)).astype(np.float32)
    W = R.standard_normal((nin + u, 4 * u)).astype(np.float32) * .4
    b = R.standard_normal(4 * u).astype(np.float32) * .1
    y, st, _ = lstm.seq_fwd(x.astype(np.float64), [(W.astype(np.float64), b.astype(np.float64))])
    m = torch.nn.LSTM(nin, u, batch_first=True)
    perm = np.concatenate([np.arange(0, u), np.arange(2 * u, 3 * u), np.arange(u, 2 * u), np.arange(3 * u, 4 * u)])
    with torch.no_grad():          # TF i,ci,f,o -> torch i,f,g,o   
        m.weight_ih_l0.copy_(torch.tensor(W[:nin].T[perm]))
        m.weight_hh_l0.copy_(torch.tensor(W[nin:].T[perm]))
        m.bias_ih_l0.copy_(torch.tensor(b[perm]))
        m.bias_hh_l0.zero_()
        yt, (hn, cn) = m(torch.tensor(x))
    assert np.allclose(y, yt.numpy(), atol=2e-6)
    assert np.allclose(st[0][0], cn[0].numpy(), atol=2e-6) and np.allclose(st[0][1], hn[0].numpy(), atol=2e-6)


def test_K14_lengths_semantics():
    B, T, nin, u = 3, 5, 2, 3
    x = R.standard_normal((B, T, nin))
    layers = [(R.standard_normal((nin + u, 4 * u)) * .5, np.zeros(4 * u))]
    lengths = np.array([5, 2, 3])
    y_dec, st_dec, _ = lstm.seq_fwd(x, layers, lengths=lengths, flavour='decode')
    y_dyn, st_dyn, _ = lstm.seq_fwd(x, layers, lengths=lengths, flavour='dynamic_rnn')
    m = S.sequence_mask(lengths, T)
    assert np.allclose(y_dec[m], y_dyn[m])                 # identical on valid rows
    assert (y_dyn[~m] == 0).all() and not (y_dec[~m] == 0).all()
    # dynamic_rnn final state == state at the last valid step
    y2, st2, _ = lstm.seq_fwd(x[1:2, :2], layers)
    assert np.allclose(st_dyn[0][1][1], st2[0][1][0]) and not np.allclose(st_dec[0][1][1], st2[0][1][0])
    # flatten order: b-major then t
    flat = S.flatten_maybe_padded_sequences(y_dyn, lengths)
    assert flat.shape[0] == 10 and np.allclose(flat[5], y_dyn[1, 0]) and np.allclose(flat[7], y_dyn[2, 0])


def test_dropout_semantics():
    x = np.ones((2, 4), np.float32)
    u = np.array([[0.0, 0.09, 0.11, 0.999]] * 2, np.float32)
    y, keep = S.dropout_output(x, 0.9, u)
    assert np.array_equal(keep[0], [0, 0, 1, 1]) and np.allclose(y[0, 2], 1 / np.float32(0.9))


# ------------------------------ optimiser ---------------------------------- #
def test_K15_tf_adam_one_step():
    th, g = np.array([1.0, -2.0]), np.array([0.5, -0.25])
    t2, m, v = S.adam_tf_step(th, g, np.zeros(2), np.zeros(2), 1, 0.01)
    # t=1: m=.1g, v=.001g^2, lr_t = lr*sqrt(.001)/.1
    ref = th - 0.01 * np.sqrt(0.001) / 0.1 * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-4)
    assert np.allclose(t2, ref, rtol=1e-14)


def test_K16_clip_scale():
    g = [np.array([3.0, 4.0]), np.array([12.0])]
    c, gn = S.clip_by_global_norm(g, 5.0)
    assert np.isclose(gn, 13.0) and np.allclose(c[0], np.array([3, 4]) * 5 / 13)
    c2, _ = S.clip_by_global_norm(g, 50.0)
    assert np.allclose(c2[0], g[0])


# ------------------------------ plumbing ----------------------------------- #
def test_K17_joint_flatten_index():
    B, T, P, M = 2, 3, 4, 5
    x = R.random((B, T, P, M))
    inp, tgt = G.joint_inputs(x)
    assert inp.shape == (B, T, P * M)
    assert (inp[:, 0] == 0).all() and np.array_equal(inp[:, 1:], tgt[:, :-1])
    assert tgt[1, 2, 3 * M + 2] == x[1, 2, 3, 2]
    tr = G.per_track_inputs(x)
    assert len(tr) == M and tr[2].shape == (B, T + 1, P) and tr[2][1, 3, 1] == x[1, 2, 1, 2]


def test_K18_training_windows():
    w = G.training_windows(10, [10, 3, 7], 4)
    assert [(a, b) for a, b, _, _ in w] == [(0, 4), (4, 8), (8, 10)]
    assert list(w[0][2]) == [0, 1, 2] and list(w[0][3]) == [4, 3, 4]
    assert list(w[1][2]) == [0, 2] and list(w[1][3]) == [4, 3]
    assert list(w[2][2]) == [0] and list(w[2][3]) == [2]


# --------------- two independent restatements agree (float64/float32) ------- #
@pytest.mark.parametrize("tracks,ragged", [(1, False), (1, True), (3, False)])
def test_numpy_vs_torch_autograd_train_step(tracks, ragged):
    B, T, P = 3, 5, 4
    M = tracks if tracks > 1 else 2
    x = (R.random((B, T, P, M)) < .3).astype(np.float64)
    inp, tgt = G.joint_inputs(x)
    D = P * M if tracks == 1 else P
    p = G.init_rnn_nade(3, P * M, D, 6, [8, 5], np.float64, tracks)
    lengths = np.array([5, 2, 4]) if ragged else None
    du = G.dropout_uniforms(11, B, T, [8, 5])
    fw = G.rnn_nade_forward(inp, tgt, lengths, p, 0.9, du, tracks)
    g = G.rnn_nade_backward(fw, p, tracks)
    P_t = TR.to_torch(p, dtype=torch.float64)
    loss, nlls, conds = TR.rnn_nade_loss(torch.tensor(inp), torch.tensor(tgt), lengths, P_t, 0.9,
                                         [torch.tensor(a) for a in du], tracks)
    loss.backward()
    assert np.isclose(fw['loss'], loss.item(), rtol=1e-12)
    for a, b in zip(G.flat_grads(g), TR.flat_params(P_t)):
        assert np.allclose(a, b.grad.numpy(), rtol=1e-9, atol=1e-12)
    for m in range(tracks):
        assert np.allclose(fw['cond_p'][m], conds[m].detach().numpy(), rtol=1e-12)
    # optimiser step agrees too
    opt = G.new_opt(G.flat_params(p))
    gn = G.apply_clip_adam(G.flat_params(p), G.flat_grads(g), opt)
    topt = TR.TFAdam(TR.flat_params(P_t))
    gn_t = topt.step()
    assert np.isclose(gn, gn_t, rtol=1e-10)
    for a, b in zip(G.flat_params(p), TR.flat_params(P_t)):
        assert np.allclose(a, b.detach().numpy(), rtol=1e-9, atol=1e-12)


def test_numpy_vs_torch_rbm_grads():
    B, T, D, Hn = 2, 4, 6, 5
    x = (R.random((B, T, D, 1)) < .4).astype(np.float64)
    inp, tgt = G.joint_inputs(x)
    p = G.init_rnn_rbm(5, D, D, Hn, [7, 4], np.float64)
    p['bh'] += .1; p['bv'] -= .2
    fw = G.rnn_rbm_forward(inp, tgt, None, p, 3, seed=9)
    g = G.rnn_rbm_backward(fw, p)
    P_t = TR.to_torch(p, dtype=torch.float64)
    loss = TR.rnn_rbm_loss(torch.tensor(inp), torch.tensor(tgt), torch.tensor(fw['v_sample']), P_t)
    loss.backward()
    assert np.isclose(fw['loss'], loss.item(), rtol=1e-12)
    for k in ['W', 'bh', 'bv', 'Wuh', 'Wuv']:
        assert np.allclose(g[k], P_t[k].grad.numpy(), rtol=1e-9, atol=1e-12), k
    for (dW, db), (Wt, bt) in zip(g['lstm'], P_t['lstm']):
        assert np.allclose(dW, Wt.grad.numpy(), rtol=1e-9, atol=1e-12)
        assert np.allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_generate_scan_shapes_and_determinism():
    p = G.init_rnn_nade(3, 8, 8, 6, [8, 5], np.float32)
    intro = (R.random((2, 3, 8)) < .3).astype(np.float32)
    s1 = G.rnn_nade_generate(intro, 4, p, seed=5)
    s2 = G.rnn_nade_generate(intro, 4, p, seed=5)
    s3 = G.rnn_nade_generate(intro[1:], 4, p, seed=5, row0=1)
    assert s1.shape == (2, 4, 8) and np.array_equal(s1, s2)
    assert np.array_equal(s1[1:], s3)          # row-keyed RNG: independent of batch split


def test_det_checker_lstm_step_and_dense_against_float64():
    """The deterministic float32 checker of the sampling path (oracle/det_ref.c: quartered fmaf chains, det_sigmoid / det_tanh) against the
    float64 restatement of the same cell (tf_semantics.lstm_block_cell) and Dense: within float32 rounding, for widths that exercise the
    quarter boundaries (K not a multiple of 8, K > 1024: two chunks) -- and the closed forms W = 0 (K12) and a single non-zero input."""
    from oracle import det, tf_semantics as S
    rng = np.random.default_rng(12)
    for n_in, u in [(5, 32), (13, 64), (440, 512), (601, 512)]:
        B = 3
        W = (rng.standard_normal((n_in + u, 4 * u)) * 0.1).astype(np.float32)
        b = (rng.standard_normal(4 * u) * 0.1).astype(np.float32)
        x = rng.standard_normal((B, n_in)).astype(np.float32)
        c0 = rng.standard_normal((B, u)).astype(np.float32)
        h0 = np.tanh(rng.standard_normal((B, u))).astype(np.float32)
        _, st = det.lstm_step(x, [(c0, h0)], [(W, b)])
        h64, c64, _ = S.lstm_block_cell(x.astype(np.float64), c0.astype(np.float64), h0.astype(np.float64), W.astype(np.float64), b.astype(np.float64))
        assert np.abs(st[0][0] - c64).max() < 5e-6 and np.abs(st[0][1] - h64).max() < 5e-6, (n_in, u)
        out = det.dense(st[0][1], W[:u, :40].copy(), b[:40].copy())
        assert np.abs(out - (st[0][1].astype(np.float64) @ W[:u, :40].astype(np.float64) + b[:40])).max() < 5e-6
    # K12: W = 0, b = 0 -> gates 1/2, ci = 0: c = c_prev / 2, h = tanh(c) / 2; zero state stays zero
    u, n_in = 32, 7
    z = det.lstm_step(np.ones((2, n_in), np.float32), None, [(np.zeros((n_in + u, 4 * u), np.float32), np.zeros(4 * u, np.float32))])[1][0]
    assert not z[0].any() and not z[1].any()
    c0 = np.full((2, u), 0.8, np.float32)
    _, st = det.lstm_step(np.ones((2, n_in), np.float32), [(c0, np.zeros((2, u), np.float32))], [(np.zeros((n_in + u, 4 * u), np.float32), np.zeros(4 * u, np.float32))])
    assert np.allclose(st[0][0], 0.4, atol=1e-7) and np.allclose(st[0][1], np.tanh(0.4) / 2, atol=1e-6)
    # one non-zero input column: the sum is that single product whatever quarter the column falls into
    W = rng.standard_normal((1100, 8)).astype(np.float32)
    for k in (0, 137, 275, 549, 1023, 1024, 1099):
        x = np.zeros((1, 1100), np.float32); x[0, k] = 1.5
        assert np.array_equal(det.dense(x, W), (np.float32(1.5) * W[k])[None, :])
