"""End-to-end parity of the Generator plugin classes (RnnNade, RnnMultiNADE, RnnRBM, encoders)
against the CPU oracle: forward metrics, every gradient, and the clipped TF-Adam step.

fp32 path: 1e-4 relative (the BASELINE.json gate).  bf16 path: its own error is reported against
the fp32 oracle with a loose bound (bf16 carries 8 significant bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import generators as G, nade as onade, rbm as orbm, philox, det   # noqa: E402
from oracle import tf_semantics as S   # noqa: E402

DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


def load_nade_params(gen, p):
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W.astype(np.float32)))
        s[f"rnn/cell_{l}/bias"].copy_(dev(b.astype(np.float32)))
    s["nade/w_enc"].copy_(dev(np.stack(p['w_enc']).astype(np.float32)))
    s["nade/w_dec"].copy_(dev(np.stack(p['w_dec']).astype(np.float32)))
    s["dense/kernel"].copy_(dev(p['fc_k'].astype(np.float32)))
    s["dense/bias"].copy_(dev(p['fc_b'].astype(np.float32)))


def oracle_grad_list(g):
    out = []
    for W, b in g['lstm']:
        out += [W, b]
    out += [np.stack(g['w_enc']), np.stack(g['w_dec']), g['fc_k'], g['fc_b']]
    return out


def make_batch(B, T, P, M, seed, rho=0.25):
    R = np.random.default_rng(seed)
    return (R.random((B, T, P, M)) < rho).astype(np.uint8)


@pytest.mark.parametrize("precision,ragged,units", [("fp32", False, [32, 64]), ("fp32", True, [32, 64]), ("bf16", True, [32, 64]),
                                                    ("bf16", False, [128, 128])])          # last: the fused three-stage launches
def test_rnn_nade_joint_train_step(precision, ragged, units):
    from multinn_amd import RnnNade, AdamOptimizer
    B, T, P, M, Hn = 6, 5, 4, 3, 20
    D = P * M
    x = make_batch(B, T, P, M, 1)
    lengths = np.array([5, 2, 4, 5, 1, 3], np.int32) if ragged else None
    p = G.init_rnn_nade(3, D, D, Hn, units, np.float64)
    for W, b in p['lstm']:
        b += 0.05
    p['fc_b'] += 0.02
    gen = RnnNade(D, Hn, units, keep_prob=0.9, precision=precision, seed=23)
    gen.build_pianoroll(dev(x), None if lengths is None else dev(lengths), is_train=True, mode="train")   # materialises the store
    load_nade_params(gen, p)
    gen._packed_step = -1
    gen.build_pianoroll(dev(x), None if lengths is None else dev(lengths), is_train=True, mode="train")
    # oracle
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    du = G.dropout_uniforms(23, B, T, units)
    fw = G.rnn_nade_forward(inp, tgt, lengths, p, 0.9, du)
    g = G.rnn_nade_backward(fw, p)
    tol = 1e-4 if precision == "fp32" else 6e-2
    assert abs(float(gen.metrics['batch/loss']) - fw['loss']) < tol * abs(fw['loss'])
    assert rel(gen.log_probs.cpu().numpy(), fw['nll'][0]) < tol
    assert np.abs(gen.cond_probs.cpu().numpy() - fw['cond_p'][0]).max() < (1e-5 if precision == "fp32" else 3e-2)
    assert gen.forward().shape == fw['cond_p'][0].shape
    gen.backward()
    names = gen.store.names()
    for name, ref in zip(names, oracle_grad_list(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        assert rel(got, ref) < tol * (1 if precision == "fp32" else 2), name
    # optimiser step (clip 5.0 + TF Adam) on the oracle's own gradients vs the device step
    opt_ref = G.new_opt(G.flat_params(p))
    gn = G.apply_clip_adam(G.flat_params(p), G.flat_grads(g), opt_ref, lr=0.01)
    gen.train(AdamOptimizer(0.01), None)
    assert abs(float(gen._grad_sumsq.sqrt()) - gn) < tol * gn
    if precision == "fp32":
        ref_params = [x_ for pair in p['lstm'] for x_ in pair] + [np.stack(p['w_enc']), np.stack(p['w_dec']), p['fc_k'], p['fc_b']]
        for name, ref in zip(names, ref_params):
            # Adam's first step moves every weight by ~lr*sign(g): compare the UPDATE, not just the value
            assert np.abs(gen.store[name].cpu().numpy().reshape(ref.shape) - ref).max() < 2e-4, name


def test_ragged_index_and_row_moves_vs_numpy():
    """ops.ragged_index (the compaction of a ragged window, built on the device) against NumPy: valid rows first in time-major order, padding
    behind them, inverse permutation, header; lengths of 0, T and beyond T, B not a multiple of the kernel's chunk.  Then the two row movers:
    rows_gather16 (+ its transposed copy, zeros behind the valid rows) and rows_scatter_f32 (zeros on padding rows)."""
    from multinn_amd import ops
    rng = np.random.default_rng(3)
    for B, T in [(5, 7), (300, 9), (1024, 4), (64, 33)]:
        ln = rng.integers(0, T + 3, B).astype(np.int32)
        ln[0], ln[-1] = 0, T
        lc = np.clip(ln, 0, T)
        idx, inv, hdr = ops.ragged_index(dev(ln), B, T, None, 256.0)
        valid = (np.arange(T)[:, None] < lc[None, :]).reshape(-1)
        rows = np.arange(T * B)
        ref_idx = np.concatenate([rows[valid], rows[~valid]])
        nv = int(valid.sum())
        assert int(hdr[0]) == nv
        assert np.array_equal(idx.cpu().numpy(), ref_idx)
        assert np.array_equal(inv.cpu().numpy()[ref_idx], np.arange(T * B))
        hf = hdr.view(torch.float32).cpu().numpy()
        assert hf[1] == np.float32(1.0) / np.float32(max(nv, 1)) and hf[2] == 2.0 ** round(np.log2(256.0 * max(nv, 1))) and hf[3] == 1.0 / hf[2]
        # an all-reduced total from the caller replaces the local count in the header's scale words
        _, _, h2 = ops.ragged_index(dev(ln), B, T, torch.tensor([3.0 * max(nv, 1)], device=DEV), 0.0)
        assert float(h2.view(torch.float32)[1]) == np.float32(1.0) / np.float32(3.0 * max(nv, 1)) and float(h2.view(torch.float32)[2]) == 1.0 and int(h2[0]) == nv
        N, C = T * B, 72
        for dt in (torch.float16, torch.bfloat16):
            src = torch.randn((N, C), device=DEV).to(dt)
            dst = torch.full((N, C), 7.0, device=DEV, dtype=dt)
            Np = (N + 63) // 64 * 64
            dstT = torch.full((C, Np), 7.0, device=DEV, dtype=dt)
            ops.rows_gather16(src, idx, hdr, dst, dstT)
            ref = torch.zeros_like(src)
            ref[:nv] = src[torch.from_numpy(ref_idx[:nv]).to(DEV)]
            assert torch.equal(dst, ref) and torch.equal(dstT[:, :N], ref.t())
        g = torch.randn((N, 8), device=DEV)
        out = torch.full((N, 8), 7.0, device=DEV)
        ops.rows_scatter_f32(g, inv, hdr, out)
        ref = torch.zeros_like(g)
        ref[torch.from_numpy(ref_idx[:nv]).to(DEV)] = g[:nv]
        assert torch.equal(out, ref)


@pytest.mark.parametrize("precision,Hn,units,B,T", [("fp16", 256, [128, 128], 96, 12), ("bf16", 20, [32, 64], 6, 5), ("fp16", 256, [512, 256], 256, 6)])
def test_ragged_window_compacted_rows_equal_the_padded_path(precision, Hn, units, B, T):
    """A ragged window with Dense + NADE on its valid rows only (the reference drops padded rows: utils/sequences.py:6-37) against the
    same step with the padding rows kept at weight 0 (`ragged_compact = False`): loss, API-order NLL rows and conditionals, every gradient --
    equal up to the summation order of the row sums.  Shapes: the split-operand matrix-core forward with its workgroups of padding skipped,
    the f32 vector forward (Hn = 20), and the row-parallel / cluster recurrence path (B = 256)."""
    from multinn_amd import RnnNade
    P, M = 8, 2
    D = P * M
    x = dev(make_batch(B, T, P, M, 11, rho=0.1))
    rng = np.random.default_rng(5)
    ln = rng.integers(1, T + 1, B).astype(np.int32)
    ln[0] = T
    ln[B // 2:] = np.minimum(ln[B // 2:], T // 2)             # whole workgroups of padding at the late timesteps
    res = []
    for compact in (True, False):
        gen = RnnNade(D, Hn, units, keep_prob=0.9, precision=precision, seed=23)
        gen._materialize(D)
        gen.ragged_compact = compact
        gen._stack.rowpar_min_batch = 32
        gen.build_pianoroll(x, dev(ln), is_train=True, mode="train")
        assert (gen._ctx.get("compact") is not None) == compact
        loss, nll, cp = float(gen.metrics["batch/loss"]), gen.log_probs.clone(), gen.cond_probs.clone()
        gen.backward()
        gen.check()
        res.append((loss, nll, cp, gen.store.grad.clone()))
    (la, na, ca, ga), (lb, nb_, cb, gb) = res
    assert na.shape == nb_.shape == (int(ln.sum()),) and ca.shape == cb.shape
    assert abs(la - lb) < 1e-6 * abs(lb)
    assert float((na - nb_).abs().max()) <= 1e-6 * float(nb_.abs().max()) and float((ca - cb).abs().max()) <= 1e-6
    assert bool(torch.isfinite(ga).all())
    assert float((ga - gb).abs().max()) <= 2e-4 * float(gb.abs().max()), float((ga - gb).abs().max()) / float(gb.abs().max())


def test_ragged_window_through_build_compacts_multinade_rows():
    """`build(x, y, lengths)` of a generator that trains on encoder outputs (composer mode: RnnMultiNADE, one NADE per track on one Dense
    output) runs Dense + NADE on the valid rows only, like `build_pianoroll` does for the joint mode: against the padded path
    (`ragged_compact = False`) -- loss, API-order per-track NLL rows and conditionals, every gradient."""
    from multinn_amd import RnnMultiNADE
    B, T, E, Mt, Hn = 96, 10, 12, 3, 256
    rng = np.random.default_rng(8)
    codes = torch.from_numpy((rng.random((B, T + 1, E * Mt)) < 0.4).astype(np.uint8)).to(DEV)
    x, y = codes[:, :-1].contiguous(), codes[:, 1:].contiguous()
    ln = rng.integers(1, T + 1, B).astype(np.int32)
    ln[0] = T
    ln[B // 2:] = np.minimum(ln[B // 2:], T // 2)
    res = []
    for compact in (True, False):
        gen = RnnMultiNADE(E, Hn, [128, 128], tracks=["a", "b", "c"], keep_prob=0.9, precision="fp16", seed=23)
        gen._materialize(E * Mt)
        gen.ragged_compact = compact
        gen.build(x, y, dev(ln), True, "train")
        assert (gen._ctx.get("compact") is not None) == compact
        loss = float(gen.metrics["batch/loss"])
        nll = [t.clone() for t in gen.log_probs]
        cp = [t.clone() for t in gen.cond_probs]
        gen.backward()
        gen.check()
        res.append((loss, nll, cp, gen.store.grad.clone()))
    (la, na, ca, ga), (lb, nb_, cb, gb) = res
    assert abs(la - lb) < 1e-6 * abs(lb)
    for a, b in zip(na, nb_):
        assert a.shape == (int(ln.sum()),) and float((a - b).abs().max()) <= 1e-6 * float(b.abs().max())
    for a, b in zip(ca, cb):
        assert float((a - b).abs().max()) <= 1e-6
    assert bool(torch.isfinite(ga).all()) and float((ga - gb).abs().max()) <= 2e-4 * float(gb.abs().max())


def test_ragged_step_is_captured_once_for_any_lengths():
    """graphed_train_step(lengths=...): ONE captured hipGraph serves every later (x, lengths) -- the compaction index, the valid-row count,
    1 / n_valid and the f16 loss scale are computed on the device inside the graph.  Replays on three different length vectors (one of them
    full-length, one nearly empty) against eager steps of a twin generator on the same data: same losses, same weights after the steps."""
    from multinn_amd import RnnNade, AdamOptimizer
    B, T, P, M, Hn, units = 64, 10, 8, 2, 256, [128, 128]
    D = P * M
    rng = np.random.default_rng(9)
    xs = [dev(make_batch(B, T, P, M, 20 + i, rho=0.1)) for i in range(3)]
    lens = [rng.integers(1, T + 1, B).astype(np.int32), np.full(B, T, np.int32), np.ones(B, np.int32)]
    lens[2][0] = 3
    a = RnnNade(D, Hn, units, keep_prob=0.9, precision="fp16", seed=23)
    b = RnnNade(D, Hn, units, keep_prob=0.9, precision="fp16", seed=23)
    a._materialize(D)
    b._materialize(D)
    b.store.theta.copy_(a.store.theta)
    oa, ob = AdamOptimizer(0.01), AdamOptimizer(0.01)
    run = a.graphed_train_step(xs[0], oa, warmup=0, lengths=dev(lens[0]))
    assert run.ragged
    for x, ln in zip(xs, lens):
        la = float(run(x, dev(ln)))
        lb = float(b.train_step(x, dev(ln), ob))
        assert abs(la - lb) < 1e-5 * abs(lb), (la, lb)
    a.check()
    b.check()
    assert float((a.store.theta - b.store.theta).abs().max()) < 2e-4


def test_fp16_ragged_lengths_loss_scale_and_overflow_guard():
    """precision='fp16' on a RAGGED window -- one long song and many rows of length 1, N / n_valid = 8 -- against the float64 oracle: the loss
    scale of the backward pass is derived from the number of VALID rows (a scale taken from B*T would multiply every gradient seed by
    N / n_valid and push dz past f16's 65504), so every gradient stays finite and within the f16 bounds.  Then the guard: a gradient norm
    that is not finite skips the optimiser step ON THE DEVICE (theta, m, v untouched) and `check()` raises."""
    from multinn_amd import RnnNade, AdamOptimizer
    B, T, P, M, Hn, units = 32, 16, 8, 2, 256, [128, 128]
    D = P * M
    x = make_batch(B, T, P, M, 5)
    lengths = np.ones(B, np.int32)
    lengths[0] = T
    lengths[1:17] = 3
    n_valid = int(lengths.sum())
    assert B * T / n_valid > 6
    p = G.init_rnn_nade(7, D, D, Hn, units, np.float64)
    for W, b in p['lstm']:
        b += 0.05
    gen = RnnNade(D, Hn, units, keep_prob=0.9, precision="fp16", seed=23)
    gen._materialize(D)
    load_nade_params(gen, p)
    gen._packed_step = -1
    gen.build_pianoroll(dev(x), dev(lengths), is_train=True, mode="train")
    # (the compacted ragged path keeps the row count and the scale ON THE DEVICE: ops.ragged_index's header)
    hdr = gen._ctx["compact"]["hdr"]
    assert int(hdr[0]) == n_valid and float(hdr.view(torch.float32)[2]) == 2.0 ** round(np.log2(256.0 * n_valid))
    assert float(hdr.view(torch.float32)[1]) == np.float32(1.0) / np.float32(n_valid) and float(gen._ctx["ls"]) == 1.0 / float(hdr.view(torch.float32)[2])
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    fw = G.rnn_nade_forward(inp, tgt, lengths, p, 0.9, G.dropout_uniforms(23, B, T, units))
    g = G.rnn_nade_backward(fw, p)
    assert abs(float(gen.metrics['batch/loss']) - fw['loss']) < 1e-4 * abs(fw['loss'])
    gen.backward()
    assert bool(torch.isfinite(gen.store.grad).all())
    for name, ref in zip(gen.store.names(), oracle_grad_list(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        assert rel(got, ref) < 3e-3, (name, rel(got, ref))
    opt = AdamOptimizer(0.01)
    gen.train(opt, None)
    gen.check()                                         # a finite norm: the step was applied, nothing to report
    # the guard: poison the gradient, the device skips the update and the host check raises
    th, m_, v_ = gen.store.theta.clone(), gen.store.m.clone(), gen.store.v.clone()
    from multinn_amd.training import compute_gradients
    gen.store.grad[3] = float("inf")
    compute_gradients(opt, gen.store, gen.clip_norm, None)
    assert torch.equal(gen.store.theta, th) and torch.equal(gen.store.m, m_) and torch.equal(gen.store.v, v_)
    with pytest.raises(FloatingPointError):
        gen.check()
    gen.check()                                         # the counter was cleared by the raise


@pytest.mark.parametrize("graphed", [False, True])
def test_fp16_overflow_lowers_the_dynamic_loss_scale_and_training_goes_on(graphed):
    """precision "fp16" with a loss scale that is far too large (loss_scale_rows x 2^14: the backward pass overflows f16): the device skips
    those steps AND halves the store's dynamic multiplier each time (ParamStore.ls_dyn, mnn_step_increment) until the gradient is finite; from
    then on steps are applied and the loss falls.  With a static scale every step would be skipped for good (seen at the bench shape after
    ~270 steps at lr 0.01).  Eager and as replays of ONE captured graph (the multiplier is read on the device).  check(): a training loop
    passes tolerate_overflow and gets a warning; the strict form still raises."""
    from multinn_amd import RnnNade, AdamOptimizer
    B, T, P, M = 64, 16, 8, 2
    x = dev(make_batch(B, T, P, M, 5, rho=0.1))
    gen = RnnNade(P * M, 256, [128, 128], keep_prob=0.9, precision="fp16", seed=23)
    gen._materialize(P * M)
    gen._stack.loss_scale_rows = 256.0 * 2.0 ** 14
    opt = AdamOptimizer(0.01)
    step = gen.graphed_train_step(x, opt, warmup=0) if graphed else (lambda: gen.train_step(x, None, opt))
    th0 = gen.store.theta.clone()
    losses = []
    for _ in range(40):
        step()
        losses.append(float(gen.metrics["batch/loss"]))
    m, inv = gen.store.ls_dyn.tolist()
    skipped, applied = int(gen.store.skipped), int(gen.store.step_dev)
    print(f"\n[dynamic loss scale, graphed={graphed}] multiplier {m:g}, skipped {skipped}, applied {applied}, loss {losses[0]:.3f} -> {losses[-1]:.3f}")
    assert 0.0 < m < 1.0 and m * inv == 1.0 and 2 <= skipped <= 20 and applied == 40 - skipped
    assert not torch.equal(gen.store.theta, th0) and bool(torch.isfinite(gen.store.theta).all())
    assert losses[-1] < 0.8 * losses[0]
    with pytest.warns(UserWarning, match="loss scale multiplier"):
        gen.check(tolerate_overflow=True)
    gen.store.skipped.fill_(1)
    with pytest.raises(FloatingPointError):
        gen.check()


@pytest.mark.parametrize("tracks", [1, 3])
def test_rnn_nade_internal_bias(tracks):
    """internal_bias=True (nade.py:69-87, rnn_nade.py:245-251): b_enc / b_dec of the NADE(s) are added to the Dense outputs.  The oracle gets
    the same model with the internal biases folded into its Dense bias: loss, NLL and every gradient must agree (fp32, 1e-4); the internal
    biases' gradients are the Dense bias gradient's two blocks, they are variables of the model (order rnn, nade, dense) and they move."""
    from multinn_amd import RnnNade, RnnMultiNADE, AdamOptimizer
    B, T, Dm, Hn, units = 5, 4, 6, 12, [32, 32]
    R = np.random.default_rng(7 + tracks)
    names = [f"t{m}" for m in range(tracks)]
    if tracks == 1:
        x = make_batch(B, T, Dm, 1, 3)
        inp, tgt = G.joint_inputs(x.astype(np.float64))
        gen = RnnNade(Dm, Hn, units, keep_prob=1.0, internal_bias=True, precision="fp32", seed=23)
        p = G.init_rnn_nade(5, Dm, Dm, Hn, units, np.float64)
    else:
        inp = (R.random((B, T, 10)) < 0.3).astype(np.float64)
        tgt = (R.random((B, T, Dm * tracks)) < 0.3).astype(np.float64)
        gen = RnnMultiNADE(Dm, Hn, units, names, keep_prob=1.0, internal_bias=True, precision="fp32", seed=23)
        p = G.init_rnn_nade(5, 10, Dm, Hn, units, np.float64, tracks=tracks)
    gen.build(dev(inp), dev(tgt), None, is_train=True, mode="train")
    assert gen.store.names()[-4:] == ["nade/b_enc", "nade/b_dec", "dense/kernel", "dense/bias"]
    load_nade_params(gen, p)
    b_int = R.standard_normal(tracks * (Hn + Dm)) * 0.2
    gen.store["nade/b_enc"].copy_(dev(b_int[:tracks * Hn].reshape(tracks, Hn).astype(np.float32)))
    gen.store["nade/b_dec"].copy_(dev(b_int[tracks * Hn:].reshape(tracks, Dm).astype(np.float32)))
    gen._packed_step = -1
    gen.build(dev(inp), dev(tgt), None, is_train=True, mode="train")
    import copy
    pf = copy.deepcopy(p)
    pf['fc_b'] = p['fc_b'] + b_int                                   # the same model, biases folded
    fw = G.rnn_nade_forward(inp, tgt, None, pf, 1.0, None, tracks=tracks)
    g = G.rnn_nade_backward(fw, pf, tracks=tracks)
    assert abs(float(gen.metrics['batch/loss']) - fw['loss']) < 1e-4 * abs(fw['loss'])
    gen.backward()
    gv = gen.store.gviews
    ref = oracle_grad_list(g)
    plain = [n for n in gen.store.names() if n not in ("nade/b_enc", "nade/b_dec")]
    for name, r_ in zip(plain, ref):
        assert rel(gv[name].cpu().numpy().reshape(r_.shape), r_) < 1e-4, name
    assert rel(gv["nade/b_enc"].cpu().numpy().ravel(), g['fc_b'][:tracks * Hn]) < 1e-4
    assert rel(gv["nade/b_dec"].cpu().numpy().ravel(), g['fc_b'][tracks * Hn:]) < 1e-4
    before = gen.store["nade/b_dec"].clone()
    gen.train(AdamOptimizer(0.01), None)
    assert float((gen.store["nade/b_dec"] - before).abs().max()) > 1e-3           # trained like every other variable
    # the sampling path sees the folded bias too: one step from the zero state reproduces b_dec + internal part on a zero Dense kernel
    gen.store["dense/kernel"].zero_()
    gen._packed_step = -1
    gen._ensure_packed()
    st = gen.zero_state(2)
    out = gen.single_step(torch.zeros((2, inp.shape[-1]), device=DEV), st)
    bd = out.b_dec if tracks == 1 else out.b_dec[0]
    want = gen.store["dense/bias"][tracks * Hn:tracks * Hn + Dm] + gen.store["nade/b_dec"][0]
    assert torch.allclose(bd[0], want, atol=1e-6)


def test_rnn_nade_generic_build_equals_pianoroll_path():
    from multinn_amd import RnnNade
    B, T, P, M, Hn, units = 4, 6, 4, 2, 16, [32, 32]
    D = P * M
    x = make_batch(B, T, P, M, 2)
    inp, tgt = G.joint_inputs(x)
    lengths = np.array([6, 3, 5, 2], np.int32)
    gen = RnnNade(D, Hn, units, keep_prob=1.0, precision="fp32", seed=5)
    gen.build(dev(inp), dev(tgt), dev(lengths), is_train=False, mode="eval")
    l1, n1 = float(gen.metrics['batch/loss']), gen.log_probs.clone()
    gen.build_pianoroll(dev(x), dev(lengths), is_train=False, mode="eval")
    assert abs(l1 - float(gen.metrics['batch/loss'])) < 1e-6 * abs(l1) and torch.allclose(n1, gen.log_probs, rtol=1e-6)
    assert n1.shape[0] == int(lengths.sum())
    with pytest.raises(ValueError):
        gen.build(dev(inp), dev(tgt), None, None, mode="bogus")
    with pytest.raises(RuntimeError):
        gen.train(None, None)                      # eval build keeps no backward context


def test_rnn_multinade_train_step():
    from multinn_amd import RnnMultiNADE
    B, T, E, M, Hn, units = 5, 4, 6, 3, 12, [32, 32]
    R = np.random.default_rng(4)
    enc = (R.random((B, T + 1, E * M)) < .3).astype(np.uint8)      # stacked per-track codes, track-minor (multinn_composer.py:73-87)
    enc[:, 0] = 0
    inp, tgt = enc[:, :-1], enc[:, 1:]
    p = G.init_rnn_nade(7, E * M, E, Hn, units, np.float64, tracks=M)
    gen = RnnMultiNADE(E, Hn, units, tracks=[f"t{m}" for m in range(M)], keep_prob=0.9, precision="fp32", seed=11)
    gen.build(dev(inp), dev(tgt), None, True, "train")
    load_nade_params(gen, p)
    gen._packed_step = -1
    gen.build(dev(inp), dev(tgt), None, True, "train")
    du = G.dropout_uniforms(11, B, T, units)
    fw = G.rnn_nade_forward(inp.astype(np.float64), tgt.astype(np.float64), None, p, 0.9, du, tracks=M)
    g = G.rnn_nade_backward(fw, p, tracks=M)
    assert abs(float(gen.metrics['batch/loss']) - fw['loss']) < 1e-4 * abs(fw['loss'])
    for m in range(M):
        assert rel(gen.log_probs[m].cpu().numpy(), fw['nll'][m]) < 1e-4
    gen.backward()
    for name, ref in zip(gen.store.names(), oracle_grad_list(g)):
        assert rel(gen.store.gviews[name].cpu().numpy().reshape(ref.shape), ref) < 1e-4, name
    st = gen.zero_state(3)
    assert len(st.b_enc) == M and st.b_enc[0].shape == (3, Hn) and len(st.rnn_state) == 2


def _check_autoregressive_samples(samples, intro, p, seed, tracks, D, tol=2e-5):
    """Teacher-forced verification of a sampling scan: replay the oracle on the DEVICE's own
    samples and require every draw to agree with u < p unless |u - p| is within tol."""
    from oracle import lstm as olstm
    B, steps, _ = samples.shape
    Hn = p['w_enc'][0].shape[1]
    y, state, _ = olstm.seq_fwd(intro.astype(np.float64), p['lstm'])
    out = S.dense(y[:, -1], p['fc_k'], p['fc_b'])
    n_amb = 0
    for s in range(steps):
        b_enc, b_dec = G.split_biases(out, Hn, D, tracks)
        u = philox.uniform_block(seed, philox.STREAM_NADE, np.arange(B), s, tracks * D)
        step = samples[:, s].astype(np.float64)
        per = [step] if tracks == 1 else [step.reshape(B, D, tracks)[..., m] for m in range(tracks)]
        for m in range(tracks):
            _, cp = onade.log_prob(per[m], b_enc[m], b_dec[m], p['w_enc'][m], p['w_dec'][m])
            um = u[:, m * D:(m + 1) * D]
            want = um < cp
            bad = want != (per[m] > 0.5)
            amb = np.abs(um - cp) < tol
            assert not (bad & ~amb).any(), f"step {s} track {m}: draw disagrees with u<p beyond tolerance"
            n_amb += int((bad & amb).sum())
        h, state = G.lstm_single_step(step, state, p['lstm'])
        out = S.dense(h, p['fc_k'], p['fc_b'])
    return n_amb


@pytest.mark.parametrize("tracks", [1, 3])
def test_generate_scan(tracks):
    from multinn_amd import RnnNade, RnnMultiNADE
    B, Ti, E, Hn, units, steps = 5, 4, 8, 16, [32, 32], 6
    Din = E * tracks
    R = np.random.default_rng(6)
    intro = (R.random((B, Ti, Din)) < .3).astype(np.uint8)
    p = G.init_rnn_nade(9, Din, E, Hn, units, np.float64, tracks=tracks)
    gen = RnnNade(E, Hn, units, precision="fp32", seed=31) if tracks == 1 else \
        RnnMultiNADE(E, Hn, units, tracks=list("abc"), precision="fp32", seed=31)
    gen._materialize(Din)
    load_nade_params(gen, p)
    out = gen.generate(dev(intro), steps)
    assert out.shape == (B, steps, Din) and out.dtype == torch.uint8
    # EVERY cell of the scan equals the deterministic float32 checker's (oracle/det_ref.c: LSTM steps, Dense, NADE conditionals)
    assert np.array_equal(out.cpu().numpy(), det.rnn_nade_generate(intro, steps, p, 31, tracks=tracks))
    _check_autoregressive_samples(out.cpu().numpy(), intro, p, 31, tracks, E)          # and the float64 oracle agrees wherever |u - p| > 2e-5
    assert torch.equal(out, gen.generate(dev(intro), steps))          # fixed RNG -> reproducible
    gen.row0 = 2                                                        # row-keyed RNG: a batch slice reproduces its rows
    sub = gen.generate(dev(intro[2:]), steps)
    assert torch.equal(sub, out[2:])


@pytest.mark.parametrize("precision", ["fp16", "fp32"])
def test_generate_scan_bit_exact_at_bench_size(precision):
    """BASELINE.json: "bit-exact for Bernoulli sampling indices under a fixed RNG" -- the WHOLE scan of the bench's sampling leg (72 intros of
    32 steps -> 128 generated steps, D = 440, NADE 256, LSTM [512, 256]; rnn_estimator.py:271-323) against the deterministic float32
    checker: 72 x 128 x 440 cells, every one equal.  The scan runs in f32 on the master weights whatever the training precision is."""
    from multinn_amd import RnnNade
    B, Ti, D, Hn, units, steps = 72, 32, 440, 256, [512, 256], 128
    R = np.random.default_rng(61)
    intro = (R.random((B, Ti, D)) < 0.03).astype(np.uint8)
    p = G.init_rnn_nade(62, D, D, Hn, units, np.float32)
    p['fc_b'][Hn:] += np.float32(np.log(0.05 / 0.95))                  # piano-roll-like conditionals instead of coin flips
    gen = RnnNade(D, Hn, units, precision=precision, seed=23)
    gen._materialize(D)
    load_nade_params(gen, p)
    assert gen.det_sampling
    got = gen.generate(dev(intro), steps).cpu().numpy()
    ref = det.rnn_nade_generate(intro, steps, p, 23)
    assert got.shape == ref.shape == (B, steps, D)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} cells differ; first at {np.argwhere(got != ref)[:3].tolist()}"
    assert 0.005 < got.mean() < 0.5 and got[:, 96:].any()


@pytest.mark.parametrize("kind", ["nade", "multinade", "rbm"])
def test_generate_graph_replay_equals_eager_scan(monkeypatch, kind):
    """generate() replays ONE captured hipGraph of the whole scan (common.ScanGraphs); MULTINN_GENERATE_GRAPH=0 runs the eager loop.
    Same kernels and RNG counters: the bits must agree -- also for a second input through the same graph, and after the weights
    have changed (packing is part of the captured scan)."""
    from multinn_amd import RnnNade, RnnMultiNADE, RnnRBM
    B, Ti, E, Hn, units, steps = 40, 5, 16, 32, [128, 128], 7
    tracks = 3 if kind == "multinade" else 1
    Din = E * tracks
    R = np.random.default_rng(11)
    xa = dev((R.random((B, Ti, Din)) < .3).astype(np.uint8))
    xb = dev((R.random((B, Ti, Din)) < .3).astype(np.uint8))
    if kind == "nade":
        gen = RnnNade(E, Hn, units, precision="bf16", seed=5)
    elif kind == "multinade":
        gen = RnnMultiNADE(E, Hn, units, tracks=list("abc"), precision="bf16", seed=5)
    else:
        gen = RnnRBM(E, Hn, units, k=3, precision="bf16", seed=5)
    gen._materialize(Din)

    def eager(x):
        monkeypatch.setenv("MULTINN_GENERATE_GRAPH", "0")
        out = gen.generate(x, steps)
        monkeypatch.delenv("MULTINN_GENERATE_GRAPH")
        return out

    ga, gb = gen.generate(xa, steps), gen.generate(xb, steps)
    assert len(gen._scan_graphs._cache) == 1                       # the second call replayed the first one's graph
    assert torch.equal(ga, eager(xa)) and torch.equal(gb, eager(xb)) and not torch.equal(ga, gb)
    gen.store.theta.mul_(1.5)                                      # "training happened": no recapture, the replay re-packs
    gen.store.step += 1
    gc = gen.generate(xa, steps)
    assert len(gen._scan_graphs._cache) == 1
    assert torch.equal(gc, eager(xa)) and not torch.equal(gc, ga)


@pytest.mark.parametrize("bias_mode", ["conditional", "internal"])
def test_rnn_rbm_train_step(bias_mode):
    from multinn_amd import RnnRBM
    B, T, D, Hn, units, k = 5, 4, 12, 20, [32, 32], 3
    x = make_batch(B, T, D, 1, 8)
    inp, tgt = G.joint_inputs(x)
    p = G.init_rnn_rbm(5, D, D, Hn, units, np.float64)
    p['bh'] += 0.1
    p['bv'] -= 0.2
    gen = RnnRBM(D, Hn, units, keep_prob=1.0, k=k, precision="fp32", seed=13, bias_mode=bias_mode)
    gen.build(dev(inp), dev(tgt), None, True, "train")
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W.astype(np.float32))); s[f"rnn/cell_{l}/bias"].copy_(dev(b.astype(np.float32)))
    for kk in ("W", "bh", "bv"):
        s[f"rbm/{kk}"].copy_(dev(p[kk].astype(np.float32)))
    s["Wuh"].copy_(dev(p['Wuh'].astype(np.float32))); s["Wuv"].copy_(dev(p['Wuv'].astype(np.float32)))
    gen._packed_step = -1
    gen.build(dev(inp), dev(tgt), None, True, "train")
    rows = np.array([t * 65536 + b for b in range(B) for t in range(T)])     # b-major flat order, time/batch keyed ids
    fw = G.rnn_rbm_forward(inp.astype(np.float64), tgt.astype(np.float64), None, p, k, seed=13, bias_mode=bias_mode, row_ids=rows)
    vs_dev = gen._outputs.cpu().numpy()
    agree = (vs_dev == fw['v_sample']).all(1).mean()
    assert agree >= 0.9, f"only {agree:.2f} of the Gibbs rows agree with the float64 oracle"
    # evaluate the oracle on the device's own sample so a rare u~p flip cannot hide an arithmetic error
    fw['v_sample'] = vs_dev.astype(np.float64)
    bh_u, bv_u = (fw['bh_t'], fw['bv_t']) if bias_mode == "conditional" else (p['bh'], p['bv'])
    cost, F = orbm.free_energy_cost(fw['tgt'], fw['v_sample'], p['W'], bh_u, bv_u)
    assert rel(gen.cost.cpu().numpy(), cost) < 1e-4 and rel(gen.free_energy.cpu().numpy(), F) < 1e-4
    assert abs(float(gen.metrics['batch/loss']) - cost.mean()) < 1e-4 * max(1.0, abs(cost.mean()))
    if agree == 1.0:
        assert rel(gen.reconstruction_cost.cpu().numpy(), fw['recon']) < 1e-4
    g = G.rnn_rbm_backward(fw, p)
    gen.backward()
    gv = gen.store.gviews
    assert rel(gv["rbm/W"].cpu().numpy(), g['W']) < 1e-4
    if bias_mode == "conditional":
        assert rel(gv["rbm/bh"].cpu().numpy(), g['bh']) < 1e-4 and rel(gv["rbm/bv"].cpu().numpy(), g['bv']) < 1e-4
        assert rel(gv["Wuh"].cpu().numpy(), g['Wuh']) < 1e-4 and rel(gv["Wuv"].cpu().numpy(), g['Wuv']) < 1e-4
        for l, (dW, db) in enumerate(g['lstm']):
            assert rel(gv[f"rnn/cell_{l}/kernel"].cpu().numpy(), dW) < 1e-4
            assert rel(gv[f"rnn/cell_{l}/bias"].cpu().numpy(), db) < 1e-4
    else:
        assert float(gv["Wuh"].abs().max()) == 0 and float(gv["rnn/cell_0/kernel"].abs().max()) == 0      # R3: as written
    # sampling path: sample_single uses k = rbm.k (R1)
    out = gen.generate(dev(inp[:, :2]), 3)
    assert out.shape == (B, 3, D) and out.dtype == torch.uint8


def test_rbm_cd_update_and_dbn_encoder():
    from multinn_amd import DBNEncoder
    from multinn_amd.common import RBM, ParamStore
    R = np.random.default_rng(12)
    N, D, Hn, k = 40, 16, 12, 2
    v = (R.random((N, D)) < .3).astype(np.uint8)
    store = ParamStore(torch.device(DEV))
    rbm = RBM(D, Hn, k=k)
    rbm.declare(store, torch.Generator().manual_seed(1))
    store.materialize()
    W0, bh0, bv0 = rbm.W.cpu().numpy().copy(), rbm.bh.cpu().numpy().copy(), rbm.bv.cpu().numpy().copy()
    rbm.seed = 99
    _, upd, grads = rbm.train(dev(v), lr=0.1, row0=7, sub0=0)
    rows = np.arange(7, 7 + N)
    u_h, u_v = G.gibbs_uniforms(99, rows, k, Hn, D, 0)
    u0 = philox.uniform_block(99, philox.STREAM_RBM_H, rows, k, Hn)
    uk = philox.uniform_block(99, philox.STREAM_RBM_H, rows, k + 1, Hn)
    dW, dbv, dbh = orbm.cd_update(v.astype(np.float64), W0.astype(np.float64), bh0.astype(np.float64), bv0.astype(np.float64), k, 0.1,
                                  u_h, u_v, u0, uk)
    assert rel(grads[0].cpu().numpy(), dW) < 1e-4 and rel(grads[1].cpu().numpy(), dbv) < 1e-4 and rel(grads[2].cpu().numpy(), dbh) < 1e-4
    assert rel(rbm.W.cpu().numpy(), W0 + dW) < 1e-5
    # DBN encoder: sampled codes up, reconstruction down (dbn_encoder.py:136-190)
    enc = DBNEncoder(D, [12, 8], seed=3)
    x = dev(v.reshape(4, 10, D))
    enc.build(x)
    assert len(enc.encodings) == 2 and enc.encodings[0].shape == (4, 10, 12)               # per-layer lists (dbn_encoder.py:83-97)
    assert enc.encodings[-1].shape == (4, 10, 8) and enc.encodings[-1].dtype == torch.uint8
    assert enc.dec_probs[0].shape == (4, 10, D) and float(enc.dec_probs[0].min()) > 0 and float(enc.dec_probs[0].max()) < 1
    assert enc.decodings[1].shape == (4, 10, 12) and torch.equal(enc.encode()[1], enc.encodings[-1])
    W1, W2 = [r.W.cpu().numpy() for r in enc.dbn.rbms]
    b1, b2 = [r.bh.cpu().numpy() for r in enc.dbn.rbms]
    ph1 = det.rbm_hidden(v, W1, b1)
    h1 = (philox.uniform_block(3, philox.STREAM_DBN_ENC, np.arange(N), 0, 12) < ph1).astype(np.uint8)
    ph2 = det.rbm_hidden(h1, W2, b2)
    h2 = (philox.uniform_block(3, philox.STREAM_DBN_ENC, np.arange(N), 1, 8) < ph2).astype(np.uint8)
    assert np.array_equal(enc.encodings[-1].cpu().numpy().reshape(N, 8), h2)
    assert np.array_equal(enc.enc_probs[-1].cpu().numpy().reshape(N, 8), ph2)
    assert np.array_equal(enc.encodings[0].cpu().numpy().reshape(N, 12), h1)
    assert torch.equal(enc.encode(x)[1], enc.encodings[-1])                                  # explicit encode == the built encodings
    W2_before = enc.dbn.rbms[1].W.clone()
    io, uo, mt, mu, sm = enc.train(None, 0.05, layer=1)                                      # dbn_encoder.py:192-240
    assert len(io) == 1 and {"batch/loss", "free_energy", "log_likelihood"} <= set(mt) and not torch.equal(W2_before, enc.dbn.rbms[1].W)


def test_feedback_rnn_sampling_scan():
    """C5 / A19: M per-track RnnNade generators + recurrent feedback module, teacher-forced replay check."""
    from multinn_amd import RnnNade
    from multinn_amd.feedback import FeedbackRnn, FeedbackRnnSampler
    B, Ti, P, M, Hn, F, steps = 4, 3, 8, 3, 16, 32, 5
    R = np.random.default_rng(14)
    x = (R.random((B, Ti, P, M)) < .3).astype(np.uint8)
    fb = FeedbackRnn(P * M, [64, F], precision="fp32", seed=40)
    gens, gparams, seeds = [], [], []
    for i in range(M):
        g = RnnNade(P, Hn, [32, 32], precision="fp32", seed=50 + i)
        g._materialize(P + F)
        p = G.init_rnn_nade(60 + i, P + F, P, Hn, [32, 32], np.float64)
        load_nade_params(g, p)
        gens.append(g); gparams.append(p); seeds.append(50 + i)
    fb_layers = [(fb.store[f"feedback/rnn/cell_{l}/kernel"].cpu().numpy().astype(np.float64),
                  fb.store[f"feedback/rnn/cell_{l}/bias"].cpu().numpy().astype(np.float64)) for l in range(2)]
    out = FeedbackRnnSampler(gens, fb).generate(dev(x), steps)
    assert out.shape == (B, steps, P, M) and out.dtype == torch.uint8
    got = out.cpu().numpy()
    probs, us = G.feedback_rnn_teacher_forced(x, got, gparams, fb_layers, seeds)
    bad = (us < probs) != (got > 0)
    assert not (bad & (np.abs(us - probs) > 2e-5)).any()
    # every cell equals the deterministic float32 checker's scan (oracle/det.py: feedback LSTM, M generators, Dense, NADE draws)
    assert np.array_equal(got, det.feedback_rnn_generate(x, steps, gparams, fb_layers, seeds))
    assert torch.equal(out, FeedbackRnnSampler(gens, fb).generate(dev(x), steps))


def test_feedback_scan_bit_exact_at_real_widths():
    """C5 at its real widths (default_feedback_rnn.yaml:11-13: generators [256, 256], NADE 256, feedback LSTM [256, 128], P = 88, 5 tracks):
    8 intros of 8 steps -> 64 generated steps, every cell of the 8 x 64 x 88 x 5 piano-roll equal to the deterministic checker's."""
    from multinn_amd import RnnNade
    from multinn_amd.feedback import FeedbackRnn, FeedbackRnnSampler
    B, Ti, P, M, Hn, F, steps = 8, 8, 88, 5, 256, 128, 64
    R = np.random.default_rng(71)
    x = (R.random((B, Ti, P, M)) < .05).astype(np.uint8)
    fb = FeedbackRnn(P * M, [256, F], precision="fp16", seed=40)
    gens, gparams, seeds = [], [], []
    for i in range(M):
        g = RnnNade(P, Hn, [256, 256], precision="fp16", seed=50 + i)
        g._materialize(P + F)
        p = G.init_rnn_nade(80 + i, P + F, P, Hn, [256, 256], np.float32)
        p['fc_b'][Hn:] += np.float32(np.log(0.1 / 0.9))
        load_nade_params(g, p)
        gens.append(g); gparams.append(p); seeds.append(50 + i)
    fb_layers = [(fb.store[f"feedback/rnn/cell_{l}/kernel"].cpu().numpy(), fb.store[f"feedback/rnn/cell_{l}/bias"].cpu().numpy()) for l in range(2)]
    got = FeedbackRnnSampler(gens, fb).generate(dev(x), steps).cpu().numpy()
    ref = det.feedback_rnn_generate(x, steps, gparams, fb_layers, seeds)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} cells differ; first at {np.argwhere(got != ref)[:3].tolist()}"
    assert 0.005 < got.mean() < 0.6


def test_dense_feedback_module_and_sampling_scan():
    """N4: the Dense feedback module (models/common/dnn.py, multinn_feedback.py:46-52,103-123) -- sigmoid Dense layers over the
    stacked codes, stateless -- against NumPy, and the feedback sampling scan run with it."""
    from multinn_amd import DNN, FeedbackDnn, FeedbackSampler, RnnNade
    rng = np.random.default_rng(3)
    x = rng.random((5, 7, 24)).astype(np.float32)
    net = DNN([16, 8], num_inputs=24, seed=4)
    h = x.reshape(-1, 24).astype(np.float64)
    for l in range(2):
        W = net.store[f"dnn/dense_{l}/kernel"].cpu().numpy().astype(np.float64)
        b = net.store[f"dnn/dense_{l}/bias"].cpu().numpy().astype(np.float64)
        lim = np.sqrt(6.0 / sum(W.shape))
        assert np.abs(W).max() <= lim + 1e-6 and np.abs(W).max() > 0.5 * lim and not b.any()       # Xavier-uniform kernels, zero biases
        h = 1.0 / (1.0 + np.exp(-(h @ W + b)))
    got = net(dev(x)).cpu().numpy()
    assert got.shape == (5, 7, 8) and np.abs(got.reshape(-1, 8) - h).max() < 1e-5
    P, M, B = 8, 2, 3
    fb = FeedbackDnn(P * M, [12], seed=1)
    out, st = fb.run(dev(rng.random((B, 4, P * M)).astype(np.float32)))
    assert out.shape == (B, 4, 12) and st is None
    gens = [RnnNade(P, 8, [32, 32], keep_prob=1.0, precision="fp32", seed=10 + i) for i in range(M)]
    smp = FeedbackSampler(gens, fb)
    intro = dev(make_batch(B, 5, P, M, 6))
    s1 = smp.generate(intro, 3)
    assert s1.shape == (B, 3, P, M) and s1.dtype == torch.uint8 and torch.equal(s1, smp.generate(intro, 3))


def test_driver_fit_and_checkpoints(tmp_path):
    """A13: the train.py loop (windows, ragged lengths, best/last checkpoints) on a tiny synthetic set."""
    from multinn_amd import RnnNade, AdamOptimizer
    from multinn_amd.driver import fit, TrainingStats
    R = np.random.default_rng(3)
    X = (R.random((12, 16, 8, 2)) < .15).astype(np.uint8)
    lengths = np.array([16, 16, 9, 16, 4, 16, 16, 12, 16, 16, 16, 7])
    gen = RnnNade(16, 16, [32], keep_prob=0.9, precision="fp32", seed=1)
    dirs = dict(model_dir=str(tmp_path / "best"), model_last_dir=str(tmp_path / "last"))
    cfg = dict(epochs=3, batch_size=4, piece_size=2, early_stopping=5, learning_rate=0.02)
    stats = fit(gen, AdamOptimizer(0.02), X[:8], lengths[:8], X[8:], lengths[8:], cfg, dict(evaluate_epochs=1, save_checkpoint_epochs=1),
                dirs, TrainingStats(), beat_size=4, log=lambda *_: None)
    assert stats.epoch == 3 and stats.steps == 6                      # steps count song batches, not pieces (train.py:194)
    assert (tmp_path / "best" / "rnn-nade.pt").exists() and (tmp_path / "last" / "rnn-nade.pt").exists()
    assert stats.metric_best < float("inf")


def test_driver_epoch_replays_captured_steps(monkeypatch):
    """train_epoch runs windows of a recurring shape as hipGraph replays (captured at the shape's second occurrence, nothing executed by the
    capture): the loss trajectory must be the eager loop's (MULTINN_TRAIN_GRAPH=0).  Ragged windows are captured too (their own graph beside
    the full-length one of the same shape: the compacted ragged step keeps its row counts on the device, so it serves any lengths)."""
    from multinn_amd import RnnNade, AdamOptimizer
    from multinn_amd.driver import train_epoch, LossAccumulator, TrainingStats
    R = np.random.default_rng(5)
    X = (R.random((24, 12, 8, 2)) < .2).astype(np.uint8)
    lengths = np.full(24, 12)
    lengths[5] = 7                                               # one ragged song: its late window is a ragged step
    ids = np.arange(24)

    def epoch_losses(graph):
        if not graph:
            monkeypatch.setenv("MULTINN_TRAIN_GRAPH", "0")
        gen = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=2)
        gen._materialize(16)
        opt = AdamOptimizer(0.01)
        out = []
        for _ in range(2):
            acc = LossAccumulator()
            train_epoch(gen, X, lengths, ids, 8, 4, opt, acc, TrainingStats(), lr=0.01, device=DEV)
            out.append(acc.loss())
        if not graph:
            monkeypatch.delenv("MULTINN_TRAIN_GRAPH")
        return out, gen

    lg, gg = epoch_losses(True)
    le, ge = epoch_losses(False)
    assert len(gg.__dict__.get("_step_graphs", {})) >= 1 and "_step_graphs" not in ge.__dict__
    assert any(k[-1] == "ragged" for k in gg._step_graphs) and any(k[-1] == "full" for k in gg._step_graphs), list(gg._step_graphs)
    assert np.allclose(lg, le, rtol=5e-3), (lg, le)
    assert gg.store.step == ge.store.step
    assert torch.allclose(gg.store.theta, ge.store.theta, atol=5e-3)


def test_save_load_roundtrip(tmp_path):
    from multinn_amd import RnnNade, AdamOptimizer
    x = make_batch(4, 4, 4, 2, 3)
    gen = RnnNade(8, 8, [32], keep_prob=1.0, precision="fp32", seed=1)
    gen.train_step(dev(x), None, AdamOptimizer(0.01))
    gen.save(None, str(tmp_path))
    other = RnnNade(8, 8, [32], keep_prob=1.0, precision="fp32", seed=2)
    other._materialize(8)
    assert not other.load(None, str(tmp_path / "missing"))
    assert other.load(None, str(tmp_path))
    assert torch.equal(other.store.theta, gen.store.theta) and torch.equal(other.store.m, gen.store.m) and other.store.step == 1
    l1 = float(gen.train_step(dev(x), None, AdamOptimizer(0.01)))
    l2 = float(other.train_step(dev(x), None, AdamOptimizer(0.01)))
    assert l1 == l2


def test_wavefront_pipelining_matches_sequential():
    """Layers as a wavefront on separate streams (chunked seq calls + events) give the same step."""
    from multinn_amd import RnnNade, AdamOptimizer
    from multinn_amd.generators import LstmStack
    x = make_batch(6, 40, 8, 2, 9, rho=0.2)
    a = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    b = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    a._materialize(16); b._materialize(16)
    b.store.theta.copy_(a.store.theta)
    b._stack.pipelined = True
    a._stack.persistent = b._stack.persistent = False          # this test is about the per-layer chunked form
    a._stack.fused_layers = b._stack.fused_layers = False
    opt = AdamOptimizer(0.01)
    la = [float(a.train_step(dev(x), None, opt)) for _ in range(3)]
    lb = [float(b.train_step(dev(x), None, opt)) for _ in range(3)]
    assert np.allclose(la, lb, rtol=1e-3), (la, lb)
    assert torch.allclose(a.store.theta, b.store.theta, atol=1e-3)


def test_fused_two_layer_wavefront_matches_sequential():
    """mnn_lstm2_seq_fwd/bwd (three-stage launches, lag 2) vs the per-layer sequence."""
    from multinn_amd import RnnNade, AdamOptimizer
    for T, units in [(40, [128, 128]), (7, [256, 128]), (33, [128, 256])]:
        x = make_batch(6, T, 8, 2, 9, rho=0.2)
        a = RnnNade(16, 16, units, keep_prob=0.9, precision="bf16", seed=3)
        b = RnnNade(16, 16, units, keep_prob=0.9, precision="bf16", seed=3)
        a._materialize(16); b._materialize(16)
        b.store.theta.copy_(a.store.theta)
        a._stack.fused_layers = False
        a._stack.persistent = b._stack.persistent = False
        assert b._stack.fused_layers
        opt = AdamOptimizer(0.01)
        a.build_pianoroll(dev(x), None, True, "train"); b.build_pianoroll(dev(x), None, True, "train")
        assert b._stack._fused2(6) and not a._stack._fused2(6) and not b._stack._fused2(4096)
        assert torch.allclose(a._nll_tm, b._nll_tm, rtol=1e-4)      # same step bodies; layer 2's projection is summed in another order
        a.backward(); b.backward()
        # bf16 activations: the two summation orders differ by rounding only (a wrong mask / time index would be O(1))
        for name in a.store.names():
            ga, gb = a.store.gviews[name], b.store.gviews[name]
            assert float((ga - gb).abs().max()) < 1e-2 * float(ga.abs().max()) + 1e-9, name
        cos = torch.nn.functional.cosine_similarity(a.store.grad, b.store.grad, dim=0)
        assert float(cos) > 0.99995, float(cos)
        la = [float(a.train_step(dev(x), None, opt)) for _ in range(3)]
        lb = [float(b.train_step(dev(x), None, opt)) for _ in range(3)]
        assert np.allclose(la, lb, rtol=1e-3), (la, lb)


@pytest.mark.parametrize("B,T,units,kp", [(6, 40, [128, 128], 0.9), (70, 9, [256, 128], 0.9), (33, 12, [128, 256], 1.0), (256, 24, [512, 256], 0.9),
                                          (1024, 6, [512, 256], 0.9), (600, 5, [256, 256], 0.9)])     # the last two: several row tiles per workgroup
def test_persistent_recurrence_matches_sequential(B, T, units, kp):
    """mnn_lstm2_persist_fwd/bwd (ONE launch for all T steps, register-resident weights, flag hand-offs between the
    workgroups of a row tile) vs the per-layer launch-per-step sequence: same step bodies, layer 2's projection summed
    in another order."""
    from multinn_amd import RnnNade, AdamOptimizer
    x = make_batch(B, T, 8, 2, 9, rho=0.2)
    a = RnnNade(16, 16, units, keep_prob=kp, precision="bf16", seed=3)
    b = RnnNade(16, 16, units, keep_prob=kp, precision="bf16", seed=3)
    a._materialize(16); b._materialize(16)
    b.store.theta.copy_(a.store.theta)
    a._stack.fused_layers = False
    a._stack.persistent = False
    assert b._stack.persistent
    a.build_pianoroll(dev(x), None, True, "train"); b.build_pianoroll(dev(x), None, True, "train")
    assert b._stack._persist(B) and not a._stack._persist(B)
    b._stack.check()
    assert torch.allclose(a._nll_tm, b._nll_tm, rtol=1e-4)
    nll1 = b._nll_tm.clone()
    b.build_pianoroll(dev(x), None, True, "train")
    assert torch.equal(nll1, b._nll_tm)                          # a stale hand-off would show as run-to-run noise
    a.backward(); b.backward()
    b._stack.check()
    for name in a.store.names():
        ga, gb = a.store.gviews[name], b.store.gviews[name]
        assert float((ga - gb).abs().max()) < 1e-2 * float(ga.abs().max()) + 1e-9, name
    cos = torch.nn.functional.cosine_similarity(a.store.grad, b.store.grad, dim=0)
    assert float(cos) > 0.99995, float(cos)
    opt = AdamOptimizer(0.01)
    la = [float(a.train_step(dev(x), None, opt)) for _ in range(3)]
    lb = [float(b.train_step(dev(x), None, opt)) for _ in range(3)]
    b._stack.check()
    assert np.allclose(la, lb, rtol=1e-3), (la, lb)


def test_persistent_recurrence_with_initial_state():
    """Stateful call (rnn_estimator.py:191-269 `_get_state` with an initial state): h0 is re-laid into the exchange slabs by
    pst_fill_h0_kernel, c0 read through the pointer-selected path; against the launch-per-step kernels."""
    from multinn_amd.generators import LstmStack
    from multinn_amd import RnnNade
    B, T = 70, 5
    a = RnnNade(16, 16, [256, 128], keep_prob=1.0, precision="bf16", seed=3)
    a._materialize(16)
    a._ensure_packed()
    st = a._stack
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.rand((T, B, st.ld0), device="cuda", generator=g) < 0.2).to(torch.bfloat16)
    state = [((torch.randn((B, u), device="cuda", generator=g) * 0.5), (torch.randn((B, u), device="cuda", generator=g) * 0.5).to(torch.bfloat16).float())
             for u in (256, 128)]                                             # [(c, h)] per layer
    st.persistent = True
    assert st._persist(B)
    y1, _, f1 = st.forward(x, 1.0, save=False, state0=state)
    st.check()
    st.persistent = False
    st.fused_layers = False
    y0, _, f0 = st.forward(x, 1.0, save=False, state0=state)
    assert float((y1.float() - y0.float()).abs().max()) < 2e-2 and float((y1.float() - y0.float()).abs().mean()) < 1e-3
    for (c1, h1), (c0, h0) in zip(f1, f0):
        assert float((c1 - c0).abs().max()) < 2e-2 and float((h1.float() - h0.float()).abs().max()) < 2e-2
    # and the state matters: zero state gives a different answer
    st.persistent = True
    yz, _, _ = st.forward(x, 1.0, save=False, state0=None)
    assert float((yz.float() - y1.float()).abs().max()) > 5e-2


def test_persistent_hand_offs_write_through_form_gives_the_same_bits(monkeypatch):
    """The store policy of the hand-offs (write-back inside one XCD when a row tile's workgroups share it, device-scope write-through
    otherwise) must not change a single bit: run one train-mode build + backward both ways (MNN_PERSIST_NO_LOCAL forces the second)."""
    from multinn_amd import RnnNade
    x = make_batch(96, 10, 8, 2, 13, rho=0.2)

    def run():
        g = RnnNade(16, 16, [256, 128], keep_prob=0.9, precision="bf16", seed=3)
        g._materialize(16)
        g._ensure_packed()
        g._stack.keep_debug = True
        g.build_pianoroll(dev(x), None, is_train=True, mode="train")
        g.backward()
        g._stack.check()
        assert g._stack._persist(96)
        return g._nll_tm.clone(), [t.clone() for t in g._stack._dbg_dzT]

    nll_a, dz_a = run()
    monkeypatch.setenv("MNN_PERSIST_NO_LOCAL", "1")
    nll_b, dz_b = run()
    monkeypatch.delenv("MNN_PERSIST_NO_LOCAL")
    assert torch.equal(nll_a, nll_b)
    for ta, tb in zip(dz_a, dz_b):
        assert torch.equal(ta, tb)


def test_persistent_launch_gives_up_loudly(monkeypatch):
    """A persistent launch whose status word is raised (here by the MNN_PERSIST_TEST_ABORT hook; in production by a workgroup whose
    bounded spin ran out) must drain -- every workgroup leaves at its next wait -- and the failure must stay visible to
    LstmStack.check() across later launches, which themselves run normally again."""
    from multinn_amd import _lib, RnnNade
    a = RnnNade(16, 16, [128, 128], keep_prob=1.0, precision="bf16", seed=3)
    a._materialize(16)
    a._ensure_packed()
    st = a._stack
    x = (torch.rand((6, 32, st.ld0), device=DEV) < .2).to(torch.bfloat16)
    assert st._persist(32)
    y_ok, _, _ = st.forward(x, 1.0, save=False)
    st.check()
    monkeypatch.setenv("MNN_PERSIST_TEST_ABORT", "1")
    st.forward(x, 1.0, save=False)
    monkeypatch.delenv("MNN_PERSIST_TEST_ABORT")
    with pytest.raises(_lib.MnnError, match="timed out"):
        st.check()
    y2, _, _ = st.forward(x, 1.0, save=False)                  # the next launch re-zeroes its flags and runs normally ...
    assert torch.equal(y2, y_ok)
    with pytest.raises(_lib.MnnError):                          # ... but the earlier failure stays on record
        st.check()


def test_graphed_train_step_matches_eager():
    from multinn_amd import RnnNade, AdamOptimizer
    x = make_batch(8, 6, 8, 2, 7, rho=0.2)
    a = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    b = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    b._materialize(16)
    a._materialize(16)
    b.store.theta.copy_(a.store.theta)
    opt = AdamOptimizer(0.01)
    run = b.graphed_train_step(dev(x), opt, warmup=2)          # 2 eager warm-up steps + 1 captured (not executed) step
    la = [float(a.train_step(dev(x), None, opt)) for _ in range(5)]
    lb = [float(run()) for _ in range(3)]
    # graph replays are steps 3,4,5 of the same trajectory (dropout seed and Adam step come from the device counter)
    assert np.allclose(lb, la[2:], rtol=2e-3), (la, lb)
    assert b.store.step == 5 and int(b.store.step_dev) == 5
    assert torch.allclose(a.store.theta, b.store.theta, atol=2e-3)


@pytest.mark.parametrize("kind", ["rbm", "multinade"])
def test_graphed_build_train_matches_eager(kind):
    """The generic captured step (build(x, y) + train as hipGraph replays) for the generators that train on encoder outputs: RnnRBM
    (jamming) and RnnMultiNADE (composer).  Replays are steps 3, 4, 5 of the eager trajectory: dropout masks, the CD-k Gibbs uniforms
    and the Adam step all follow the DEVICE step counter (a Gibbs seed baked into the graph would repeat the same chain -- and the
    same loss -- at every replay)."""
    from multinn_amd import RnnRBM, RnnMultiNADE, AdamOptimizer
    B, T, E, tracks = 8, 6, 12, (3 if kind == "multinade" else 1)
    D = E * tracks
    R = np.random.default_rng(4)
    seq = (R.random((B, T + 1, D)) < .25).astype(np.float32)
    x, y = dev(seq[:, :-1]), dev(seq[:, 1:])

    def make():
        if kind == "rbm":
            return RnnRBM(D, 20, [128, 128], keep_prob=0.9, k=3, precision="bf16", seed=3)
        return RnnMultiNADE(E, 16, [128, 128], tracks=list("abc"), keep_prob=0.9, precision="bf16", seed=3)

    a, b = make(), make()
    a._materialize(D); b._materialize(D)
    b.store.theta.copy_(a.store.theta)
    opt = AdamOptimizer(0.01)

    def eager_step():
        a.build(x, y, None, True, "train")
        a.train(opt, 0.01)
        return float(a._loss)

    run = b.graphed_build_train(x, y, opt, 0.01, warmup=2)
    la = [eager_step() for _ in range(5)]
    lb = [float(run()) for _ in range(3)]
    assert np.allclose(lb, la[2:], rtol=5e-3, atol=1e-4), (la, lb)
    assert len(set(np.round(lb, 6))) == 3                      # every replay is a different step
    assert b.store.step == 5 and int(b.store.step_dev) == 5
    assert torch.allclose(a.store.theta, b.store.theta, atol=3e-3)


def test_two_graph_data_parallel_step_matches_eager(monkeypatch):
    """The N>1 form of the captured step: forward+backward graph | (all-reduce) | clip+Adam graph, replayed back to back.
    Regression: with hipMemsetAsync nodes the second replay of the first graph started from non-zero hand-off flags."""
    from multinn_amd import RnnNade, AdamOptimizer
    import multinn_amd.generators as gmod
    monkeypatch.setattr(gmod, "dp_active", lambda: True)        # take the two-graph path without a process group
    x = make_batch(32, 12, 8, 2, 7, rho=0.2)
    a = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    b = RnnNade(16, 16, [128, 128], keep_prob=0.9, precision="bf16", seed=3)
    b._materialize(16)
    a._materialize(16)
    b.store.theta.copy_(a.store.theta)
    opt = AdamOptimizer(0.01)
    run = b.graphed_train_step(dev(x), opt, warmup=2)
    assert b._stack._persist(32)                                # the persistent recurrence is the path under test
    la = [float(a.train_step(dev(x), None, opt)) for _ in range(7)]
    lb = [float(run()) for _ in range(5)]
    b._stack.check()
    assert np.allclose(lb, la[2:], rtol=2e-3), (la, lb)
    assert b.store.step == 7 and int(b.store.step_dev) == 7
    assert torch.allclose(a.store.theta, b.store.theta, atol=3e-3)


def test_cond_probs_on_demand_after_a_train_mode_build():
    """The train step does not store the conditionals (only the loss needs them); `cond_probs` then runs one more decoder pass over
    the saved Dense output and must equal what an eval-mode build of the same batch returns."""
    from multinn_amd import RnnNade
    x = make_batch(6, 9, 8, 2, 11, rho=0.2)
    for prec in ("fp32", "bf16"):
        a = RnnNade(16, 256 if prec == "bf16" else 16, [128, 128], keep_prob=1.0, precision=prec, seed=3)
        a.build_pianoroll(dev(x), None, is_train=True, mode="train")
        assert a._cond_tm is None
        cp_train = a.cond_probs
        lp_train = a.log_probs
        a.build_pianoroll(dev(x), None, is_train=False, mode="eval")
        assert a._cond_tm is not None
        assert torch.equal(cp_train, a.cond_probs) and torch.allclose(lp_train, a.log_probs, rtol=1e-6, atol=1e-6)


def test_training_reduces_loss():
    from multinn_amd import RnnNade, AdamOptimizer
    x = make_batch(8, 8, 8, 2, 5, rho=0.1)
    gen = RnnNade(16, 16, [32, 32], keep_prob=0.9, precision="bf16", seed=3)
    opt = AdamOptimizer(0.01)
    losses = [float(gen.train_step(dev(x), None, opt)) for _ in range(40)]
    assert losses[-1] < 0.7 * losses[0], losses[::8]


def test_encoder_pretraining_driver_lowers_reconstruction_cost():
    """train_encoders.py:118-200 restated in driver.pretrain_encoders: greedy CD-k per layer over shuffled song windows."""
    from multinn_amd import DBNEncoder, driver
    R = np.random.default_rng(5)
    S, T, P, M = 12, 16, 24, 2
    proto = (R.random((4, P, M)) < .25)
    X = (proto[R.integers(0, 4, (S, T))] ^ (R.random((S, T, P, M)) < .02)).astype(np.uint8)
    lens = np.full(S, T, dtype=np.int64); lens[3] = 9
    encs = [DBNEncoder(P, [16, 8], k=2, seed=3 + i, track_name=f"t{i}", device=torch.device(DEV)) for i in range(M)]
    cfg = {"batch_size": 4, "piece_size": 2, "learning_rate": 0.05, "epochs": 6}
    lines = []
    hist = driver.pretrain_encoders(encs, X, lens, cfg, beat_size=4, device=DEV, log=lines.append)
    assert set(hist) == {(i, l) for i in range(M) for l in range(2)} and len(lines) == M * 2 * 6
    for i in range(M):
        c = hist[(i, 0)]
        assert all(np.isfinite(c)) and c[-1] < c[0], c
        m = encs[i].metrics
        assert {"loss", "log_likelihood", "batch/loss", "free_energy"} <= set(m)


@pytest.mark.parametrize("tracks", [1, 3])
def test_generate_scan_entry_point_equals_step_by_step_scan(tracks):
    """mnn_generate_scan (the whole scan of rnn_estimator.py:271-323 in one C-ABI call) against the same scan driven step by step
    through sample_single / single_step, and against the deterministic checker: identical samples; also with intros that are not byte
    tensors (those take the step-by-step path)."""
    from multinn_amd import RnnNade, RnnMultiNADE
    B, Ti, E, Hn, units, steps = 9, 5, 24, 32, [64, 32], 11
    Din = E * tracks
    R = np.random.default_rng(16)
    intro = (R.random((B, Ti, Din)) < .3).astype(np.uint8)
    p = G.init_rnn_nade(19, Din, E, Hn, units, np.float32, tracks=tracks)
    gen = RnnNade(E, Hn, units, precision="fp16", seed=37) if tracks == 1 else RnnMultiNADE(E, Hn, units, tracks=list("abc"), precision="fp16", seed=37)
    gen._materialize(Din)
    load_nade_params(gen, p)
    one = gen.generate(dev(intro), steps)                                  # byte intro: mnn_generate_scan
    assert gen._scan_in_one_call(dev(intro), steps) is not None
    stepwise = gen.generate(dev(intro).float(), steps)                     # float intro: sample_single / single_step per generated step
    assert gen._scan_in_one_call(dev(intro).float(), steps) is None
    assert torch.equal(one, stepwise)
    assert np.array_equal(one.cpu().numpy(), det.rnn_nade_generate(intro, steps, p, 37, tracks=tracks))


def test_rnn_rbm_generate_scan_bit_exact():
    """RnnRBM.generate (A11 with the LSTM-RBM estimator: Gibbs chain per generated step, rnn_rbm.py:283-297 with k = rbm.k) against the
    deterministic checker: every cell of the scan."""
    from multinn_amd import RnnRBM
    B, Ti, D, Hn, units, k, steps = 10, 4, 24, 40, [64, 32], 4, 9
    R = np.random.default_rng(26)
    intro = (R.random((B, Ti, D)) < .3).astype(np.uint8)
    p = G.init_rnn_rbm(27, D, D, Hn, units, np.float32)
    p['bh'] += np.float32(0.1); p['bv'] -= np.float32(0.3)
    gen = RnnRBM(D, Hn, units, k=k, precision="fp16", seed=41)
    gen._materialize(D)
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W)); s[f"rnn/cell_{l}/bias"].copy_(dev(b))
    for kk in ("W", "bh", "bv"):
        s[f"rbm/{kk}"].copy_(dev(p[kk]))
    s["Wuh"].copy_(dev(p['Wuh'])); s["Wuv"].copy_(dev(p['Wuv']))
    got = gen.generate(dev(intro), steps).cpu().numpy()
    ref = det.rnn_rbm_generate(intro, steps, p, k, 41)
    assert got.shape == ref.shape == (B, steps, D)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} cells differ; first at {np.argwhere(got != ref)[:3].tolist()}"
    assert 0.02 < got.mean() < 0.98
