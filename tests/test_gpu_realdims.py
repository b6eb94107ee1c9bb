"""The BENCHMARKED configuration against the float64 oracle at its REAL widths (D = 440 visibles, 256 NADE hidden units, LSTM
[512, 256]) -- not toy dims -- and a property test at the north-star shape [1024, 256, 88, 5].

fp32 mode (the parity mode): loss, per-row NLL and every gradient within 1e-4 relative of oracle/generators.py.
bf16 mode (the benchmarked one: persistent recurrence + matrix-core NADE forward, both asserted ON): its error against the same
float64 oracle is printed and bounded (bf16 operands carry 8 significant bits)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import generators as G   # noqa: E402

DEV = "cuda:0"
P, M, HN, UNITS = 88, 5, 256, [512, 256]
D = P * M


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / max(1e-300, np.linalg.norm(a) * np.linalg.norm(b)))


def synth(B, T, seed, rho):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((B, T, P, M)) < rho).astype(np.uint8)


def load(gen, p):
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W.astype(np.float32))); s[f"rnn/cell_{l}/bias"].copy_(dev(b.astype(np.float32)))
    s["nade/w_enc"].copy_(dev(np.stack(p['w_enc']).astype(np.float32)))
    s["nade/w_dec"].copy_(dev(np.stack(p['w_dec']).astype(np.float32)))
    s["dense/kernel"].copy_(dev(p['fc_k'].astype(np.float32)))
    s["dense/bias"].copy_(dev(p['fc_b'].astype(np.float32)))
    gen._packed_step = -1


def oracle_grads(g):
    out = []
    for W, b in g['lstm']:
        out += [W, b]
    return out + [np.stack(g['w_enc']), np.stack(g['w_dec']), g['fc_k'], g['fc_b']]


_ORACLE = {}


def oracle(B, T, rho):
    """float64 forward + backward of the joint LSTM-NADE step at real widths (a few seconds), cached per input."""
    key = (B, T, rho)
    if key not in _ORACLE:
        x = synth(B, T, 23, rho)
        p = G.init_rnn_nade(23, D, D, HN, UNITS, np.float64)
        for W, b in p['lstm']:
            b += 0.05                                   # non-zero biases: a dropped bias add would otherwise be invisible
        p['fc_b'] += 0.02
        p['fc_b'][HN:] += np.log(max(rho, 1e-3) / (1 - min(rho, 0.999)))      # conditionals near the data density, as after training
        inp, tgt = G.joint_inputs(x.astype(np.float64))
        fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, G.dropout_uniforms(23, B, T, UNITS))
        _ORACLE[key] = (x, p, fw, G.rnn_nade_backward(fw, p))
    return _ORACLE[key]


@pytest.mark.parametrize("rho", [0.03, 0.5])
@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_joint_lstm_nade_step_at_real_widths_vs_oracle(precision, rho):
    from multinn_amd import RnnNade
    B, T = 32, 8
    x, p, fw, g = oracle(B, T, rho)
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
    gen._materialize(D)
    load(gen, p)
    gen.build_pianoroll(dev(x), None, is_train=True, mode="train")
    if precision in ("bf16", "fp16"):
        assert gen._stack._persist(B, T) and gen._nade_mfma(), "the benchmarked kernels must be the ones under test"
    loss = float(gen.metrics["batch/loss"])
    nll = gen.log_probs.cpu().numpy()
    cp = gen.cond_probs.cpu().numpy()                   # train-mode build: one more decoder pass over the saved Dense output
    gen.backward()
    gen._stack.check()
    errs = {"loss": abs(loss - fw['loss']) / abs(fw['loss']), "nll": rel(nll, fw['nll'][0])}
    cosv = {}
    for name, ref in zip(gen.store.names(), oracle_grads(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        errs[name] = rel(got, ref)
        cosv[name] = cosine(got, ref)
    print(f"\n[{precision} rho={rho}] relative error vs float64 oracle at D=440, Hn=256, [512,256], B={B}, T={T}:")
    for k, v in errs.items():
        print(f"    {k:24s} {v:.3e}" + (f"   cos {cosv[k]:.6f}" if k in cosv else ""))
    if precision == "fp32":
        assert all(v < 1e-4 for v in errs.values()), errs                  # BASELINE.json: 1e-4 relative
    elif precision == "fp16":   # the BENCHMARKED mode: the quantities BASELINE.json names within 1e-4; gradients (f16 operands, 11 bits) bounded
        assert errs["loss"] < 1e-4 and errs["nll"] < 1e-4, errs
        assert all(v < 3e-3 for v in errs.values()), errs
        assert all(c > 0.999999 for c in cosv.values()), cosv
    else:       # measured on MI355X (round 2): loss 6e-6, NLL 1.2e-4, gradients 4e-4 .. 4.3e-3, cosine >= 0.99999
        assert errs["loss"] < 5e-4 and errs["nll"] < 2e-3, errs
        assert all(v < 2e-2 for v in errs.values()), errs
        assert all(c > 0.9999 for c in cosv.values()), cosv
    cp_err = np.abs(cp - fw['cond_p'][0]).max()
    print(f"    {'cond_probs (abs)':24s} {cp_err:.3e}")
    assert cp_err < {"fp32": 2e-5, "fp16": 1e-4, "bf16": 2e-2}[precision]


def test_real_width_optimiser_step_fp32_vs_oracle():
    """Clip 5.0 + TF-Adam on the oracle's gradients vs the device step at real widths: the UPDATE of every variable."""
    from multinn_amd import RnnNade, AdamOptimizer
    B, T = 32, 8
    x, p, fw, g = oracle(B, T, 0.03)
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="fp32", seed=23)
    gen._materialize(D)
    load(gen, p)
    before = {n: gen.store[n].clone() for n in gen.store.names()}
    gen.train_step(dev(x), None, AdamOptimizer(0.01))
    import copy
    p2 = copy.deepcopy(p)
    gn = G.apply_clip_adam(G.flat_params(p2), G.flat_grads(g), G.new_opt(G.flat_params(p2)), lr=0.01)
    assert abs(float(gen._grad_sumsq.sqrt()) - gn) < 1e-4 * gn
    ref_after = [a for pair in p2['lstm'] for a in pair] + [np.stack(p2['w_enc']), np.stack(p2['w_dec']), p2['fc_k'], p2['fc_b']]
    ref_before = [a for pair in p['lstm'] for a in pair] + [np.stack(p['w_enc']), np.stack(p['w_dec']), p['fc_k'], p['fc_b']]
    for name, ra, rb in zip(gen.store.names(), ref_after, ref_before):
        upd = (gen.store[name] - before[name]).cpu().numpy().reshape(ra.shape)
        # Adam's first step moves a weight by ~lr * g/(|g| + eps): compare the update where the gradient is not tiny against eps
        assert np.abs(upd - (ra - rb)).max() < 2e-4, name


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_target_shape_train_step_properties(precision):
    """North-star shape [1024, 256, 88, 5] (BASELINE.json) in the benchmarked mode (fp16) and in bf16: the step runs on the row-parallel
    persistent recurrence and the matrix-core NADE forward; loss finite and in range, forward bit-deterministic, no persistent launch gave
    up, gradients finite and aligned with the launch-per-step / f32-NADE kernels' on the same weights, and the captured step advances."""
    from multinn_amd import RnnNade, AdamOptimizer
    B, T = 1024, 256
    x = dev(synth(B, T, 23, 0.03))
    a = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
    a._materialize(D)
    a.build_pianoroll(x, None, True, "train")
    assert (a._stack._rowpar(B, T) or a._stack._persist(B, T)) and a._nade_mfma()
    la = float(a.metrics["batch/loss"])
    assert np.isfinite(la) and 40 < la < 400, la
    nll = a._nll_tm.clone()
    a.backward()
    a._stack.check()
    ga = a.store.grad.clone()
    assert bool(torch.isfinite(ga).all()) and float(ga.abs().max()) > 0
    a.build_pianoroll(x, None, True, "train")
    assert torch.equal(a._nll_tm, nll)                               # bit-deterministic forward at 262 144 rows
    b = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
    b._materialize(D)
    b.store.theta.copy_(a.store.theta)
    b._stack.persistent = False
    b._stack.rowpar = False
    b.nade_mfma = False
    b.build_pianoroll(x, None, True, "train")
    lb = float(b.metrics["batch/loss"])
    b.backward()
    assert abs(la - lb) < (1e-4 if precision == "fp16" else 3e-3) * abs(lb), (la, lb)
    cos = float(torch.nn.functional.cosine_similarity(ga, b.store.grad, dim=0))
    assert cos > (0.99999 if precision == "fp16" else 0.999), cos
    del b
    torch.cuda.empty_cache()
    opt = AdamOptimizer(0.01)
    run = a.graphed_train_step(x, opt, warmup=1)
    ls = [float(run()) for _ in range(3)]
    a._stack.check()
    assert all(np.isfinite(ls)) and ls[-1] < ls[0], ls


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
@pytest.mark.parametrize("units,B,T,max_g", [([512, 256], 64, 8, None), ([128, 128, 128], 96, 6, None), ([256], 32, 5, None),
                                             ([512, 256], 64, 6, 1), ([256], 128, 5, 1)])
def test_rowpar_recurrence_vs_oracle(units, B, T, max_g, precision, monkeypatch):
    """The row-parallel persistent recurrence (lstm_rowpar.hip: one launch per layer, weights in LDS, a wave per 32-row tile -- the form
    for B >= 512) against the float64 oracle, with its batch threshold lowered so the oracle stays quick: loss, NLL and every gradient
    within the bf16 bounds of the two-layer persistent form; it must also agree with the launch-per-step kernels on the same weights.
    Up to two row tiles per workgroup run as wave pairs (two waves per (row tile, unit tile) item); `max_g` = 1 (test hook MNN_ROWPAR_MAX_G)
    puts all row tiles of the small batch on one workgroup per unit tile: two tiles = both pair slots, four tiles = the one-wave-per-tile form."""
    from multinn_amd import RnnNade
    if max_g is not None:
        monkeypatch.setenv("MNN_ROWPAR_MAX_G", str(max_g))
    rho = 0.05
    x = synth(B, T, 29, rho)
    p = G.init_rnn_nade(31, D, D, HN, units, np.float64)
    for W, b in p['lstm']:
        b += 0.05
    p['fc_b'][HN:] += np.log(rho / (1 - rho))
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, G.dropout_uniforms(23, B, T, units))
    g = G.rnn_nade_backward(fw, p)
    gen = RnnNade(D, HN, units, keep_prob=0.9, precision=precision, seed=23)
    gen._materialize(D)
    load(gen, p)
    gen._stack.rowpar_min_batch = 32
    gen._stack.keep_debug = True
    gen.build_pianoroll(dev(x), None, is_train=True, mode="train")
    assert gen._stack._rowpar(B, T) and gen._ctx["lstm"][0].get("rowpar")
    loss = float(gen.metrics["batch/loss"])
    nll = gen.log_probs.cpu().numpy()
    gen.backward()
    gen._stack.check()
    errs = {"loss": abs(loss - fw['loss']) / abs(fw['loss']), "nll": rel(nll, fw['nll'][0])}
    cosv = {}
    for name, ref in zip(gen.store.names(), oracle_grads(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        errs[name] = rel(got, ref)
        cosv[name] = cosine(got, ref)
    print(f"\n[rowpar {precision} units={units} B={B} T={T}] relative error vs float64 oracle:")
    for k, v in errs.items():
        print(f"    {k:24s} {v:.3e}" + (f"   cos {cosv[k]:.6f}" if k in cosv else ""))
    if precision == "fp16":     # the benchmarked mode and form: BASELINE.json's 1e-4 on the loss and every row's NLL
        assert errs["loss"] < 1e-4 and errs["nll"] < 1e-4, errs
        assert all(v < 3e-3 for v in errs.values()), errs
        assert all(c > 0.999999 for c in cosv.values()), cosv
        cp_err = np.abs(gen.cond_probs.cpu().numpy() - fw['cond_p'][0]).max()
        print(f"    {'cond_probs (abs)':24s} {cp_err:.3e}")
        assert cp_err < 1e-4
    else:
        assert errs["loss"] < 5e-4 and errs["nll"] < 2e-3, errs
        assert all(v < 2e-2 for v in errs.values()), errs
        assert all(c > 0.9999 for c in cosv.values()), cosv
    # run-to-run: the recurrence's own outputs (no atomics upstream of them) are bit-stable
    nll0 = gen._nll_tm.clone()
    dz0 = [d.clone() for d in gen._stack._dbg_dzT]
    for _ in range(3):
        gen.build_pianoroll(dev(x), None, is_train=True, mode="train")
        gen.backward()
        assert torch.equal(gen._nll_tm, nll0) and all(torch.equal(a_, b_) for a_, b_ in zip(gen._stack._dbg_dzT, dz0))
    gen._stack.check()


def test_timed_kernels_long_sequence_vs_oracle(monkeypatch):
    """The TIMED kernel forms of the bench step -- the wave-pair row-parallel recurrence `lstm_rowpar_fwd2 / bwd2<512, Fp16F>` with TWO
    row tiles per workgroup for layer 1, the CU-resident recurrence `lstm_res_fwd / bwd_kernel<256, Fp16F>` for layer 2, the split-operand matrix-core NADE forward, the K-blocked dz^T weight-gradient operand -- against the float64
    oracle over T = 64 timesteps (the other oracle cases stop at T <= 9): flag epochs, the 8-chunk dz ring and the K-blocked layout are
    exercised for 64 hand-offs per row tile, at the real widths and the bench density rho = 0.03."""
    from multinn_amd import RnnNade
    monkeypatch.setenv("MNN_ROWPAR_MAX_G", "1")        # both row tiles of B = 64 on one workgroup per unit tile: both pair slots, as at B = 1024
    B, T, rho = 64, 64, 0.03
    x = synth(B, T, 29, rho)
    p = G.init_rnn_nade(31, D, D, HN, UNITS, np.float64)
    for W, b in p['lstm']:
        b += 0.05
    p['fc_b'][HN:] += np.log(rho / (1 - rho))
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, G.dropout_uniforms(23, B, T, UNITS))
    g = G.rnn_nade_backward(fw, p)
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="fp16", seed=23)
    gen._materialize(D)
    load(gen, p)
    gen._stack.rowpar_min_batch = 32
    gen.build_pianoroll(dev(x), None, is_train=True, mode="train")
    assert gen._stack._rowpar(B, T) and gen._ctx["lstm"][0].get("rowpar") and gen._nade_mfma() and gen._nade_exact()
    assert gen._stack.kblock_wgrads
    assert gen._stack._resident(1, B, T) and not gen._stack._resident(0, B, T)      # layer 2 (256 units) on the CU-resident kernels, layer 1 row-parallel
    loss = float(gen.metrics["batch/loss"])
    nll = gen.log_probs.cpu().numpy()
    cp_err = np.abs(gen.cond_probs.cpu().numpy() - fw['cond_p'][0]).max()
    gen.backward()
    gen._stack.check()
    errs = {"loss": abs(loss - fw['loss']) / abs(fw['loss']), "nll": rel(nll, fw['nll'][0])}
    cosv = {}
    for name, ref in zip(gen.store.names(), oracle_grads(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        errs[name] = rel(got, ref)
        cosv[name] = cosine(got, ref)
    print(f"\n[timed kernels fp16 B={B} T={T} rho={rho}] relative error vs float64 oracle:")
    for k, v in errs.items():
        print(f"    {k:24s} {v:.3e}" + (f"   cos {cosv[k]:.6f}" if k in cosv else ""))
    print(f"    {'cond_probs (abs)':24s} {cp_err:.3e}")
    assert errs["loss"] < 1e-4 and errs["nll"] < 1e-4 and cp_err < 1e-4, (errs, cp_err)
    assert all(v < 3e-3 for v in errs.values()), errs
    assert all(c > 0.99999 for c in cosv.values()), cosv


def test_timed_cluster_kernels_long_sequence_vs_oracle():
    """The kernel forms the bench step takes since round 5 -- layer 1 (512 units) on the CLUSTER form of the CU-resident recurrence
    (`lstm_cl_fwd / bwd_kernel<Fp16F>`: eight CUs share 32 rows, h through the XCD's L2, backward as K split + reduce-scatter of 16-bit partial
    sums), layer 2 on the CU-resident kernels, the weight-resident input-projection GEMM, the split-operand matrix-core NADE forward, the
    K-blocked dz^T operand -- against the float64 oracle over T = 8 timesteps at the real widths and the bench density (B = 256: eight
    clusters, one per XCD; 8 exchanges per cluster and direction, both exchange buffers re-used 4 times; T = 16 and T = 32 measured the same
    bounds -- the float64 oracle is what takes the time)."""
    from multinn_amd import RnnNade
    B, T, rho = 256, 8, 0.03
    x = synth(B, T, 37, rho)
    p = G.init_rnn_nade(41, D, D, HN, UNITS, np.float64)
    for W, b in p['lstm']:
        b += 0.05
    p['fc_b'][HN:] += np.log(rho / (1 - rho))
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, G.dropout_uniforms(23, B, T, UNITS))
    g = G.rnn_nade_backward(fw, p)
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="fp16", seed=23)
    gen._materialize(D)
    load(gen, p)
    gen._stack.rowpar_min_batch = 32
    gen.build_pianoroll(dev(x), None, is_train=True, mode="train")
    assert gen._stack._rowpar(B, T) and gen._ctx["lstm"][0].get("rowpar") and gen._nade_mfma() and gen._nade_exact()
    assert gen._stack.kblock_wgrads
    assert gen._stack._cluster(0, B, T) and gen._stack._resident(1, B, T) and not gen._stack._resident(0, B, T)
    loss = float(gen.metrics["batch/loss"])
    nll = gen.log_probs.cpu().numpy()
    cp_err = np.abs(gen.cond_probs.cpu().numpy() - fw['cond_p'][0]).max()
    gen.backward()
    gen._stack.check()
    errs = {"loss": abs(loss - fw['loss']) / abs(fw['loss']), "nll": rel(nll, fw['nll'][0])}
    cosv = {}
    for name, ref in zip(gen.store.names(), oracle_grads(g)):
        got = gen.store.gviews[name].cpu().numpy().reshape(ref.shape)
        errs[name] = rel(got, ref)
        cosv[name] = cosine(got, ref)
    print(f"\n[cluster kernels fp16 B={B} T={T} rho={rho}] relative error vs float64 oracle:")
    for k, v in errs.items():
        print(f"    {k:24s} {v:.3e}" + (f"   cos {cosv[k]:.6f}" if k in cosv else ""))
    print(f"    {'cond_probs (abs)':24s} {cp_err:.3e}")
    assert errs["loss"] < 1e-4 and errs["nll"] < 1e-4 and cp_err < 1e-4, (errs, cp_err)
    assert all(v < 3e-3 for v in errs.values()), errs
    assert all(c > 0.99999 for c in cosv.values()), cosv


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
def test_cluster_backward_falls_back_to_rowpar_and_agrees_over_a_long_sequence(precision, monkeypatch):
    """Two things at once, over T = 64 timesteps at the real widths (B = 256: eight clusters, 64 exchanges per cluster and direction, the production
    loss scale in fp16):
      * the cluster backward exchanges its eight partial dh sums rounded to 16 bits where the row-parallel backward accumulates the whole
        K = 2048 contraction in f32 -- the rounding compounds along the chain, so the two are compared over a LONG sequence (every gradient of the
        step, relative to its largest element: the bound recorded in DESIGN.md);
      * when the library says a cluster is not dealt onto one XCD (`mnn_lstm_cluster_bwd_ok` = 0; forced here through MNN_PERSIST_NO_LOCAL, which
        also switches the forward to its write-through hand-offs) the layer's backward runs on `lstm_rowpar_bwd` -- no error, no unwritten
        gradients, one warning."""
    import warnings
    from multinn_amd import RnnNade, ops
    from multinn_amd.generators import LstmStack
    B, T, rho = 256, 64, 0.03
    x = dev(synth(B, T, 43, rho))

    def run():
        gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
        gen._materialize(D)
        gen._stack.rowpar_min_batch = 32
        gen.build_pianoroll(x, None, is_train=True, mode="train")
        assert gen._stack._rowpar(B, T) and gen._stack._cluster(0, B, T)
        took = gen._stack._cluster_bwd(0, B, T)
        gen.backward()
        gen.check()
        return took, float(gen.metrics["batch/loss"]), gen.store.grad.clone(), {n: gen.store.gviews[n].clone() for n in gen.store.names()}

    assert ops.lstm_cluster_bwd_ok(B, 512)
    took_a, loss_a, flat_a, ga = run()
    monkeypatch.setenv("MNN_PERSIST_NO_LOCAL", "1")
    monkeypatch.setattr(LstmStack, "_cluster_bwd_warned", False, raising=False)
    assert not ops.lstm_cluster_bwd_ok(B, 512)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        took_b, loss_b, flat_b, gb = run()
    assert took_a and not took_b
    assert any("lstm_rowpar_bwd" in str(m.message) for m in w)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_a)         # same forward (the loss is summed with f32 atomics: order-dependent in the last bit)
    assert bool(torch.isfinite(flat_a).all()) and bool(torch.isfinite(flat_b).all())
    tol = 2e-3 if precision == "fp16" else 2e-2
    print(f"\n[cluster vs row-parallel backward, {precision}, T={T}] relative difference per variable:")
    for n in ga:
        d = float((ga[n] - gb[n]).abs().max()) / max(float(gb[n].abs().max()), 1e-30)
        print(f"    {n:24s} {d:.3e}")
        assert d < tol, (n, d)
    assert float(torch.nn.functional.cosine_similarity(flat_a, flat_b, dim=0)) > 0.99999


def test_persistent_forms_refuse_grids_that_cannot_be_resident():
    """Co-residency is a construction, not an assumption: the host plans size every persistent grid to at most one workgroup per CU of THIS
    device and refuse shapes whose row tiles do not fit (the caller then takes the launch-per-timestep kernels -- still device code); the C
    entry points refuse them too instead of launching a grid that could wait for workgroups that are not resident."""
    from multinn_amd import ops, RnnNade
    from multinn_amd._lib import MnnError
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert ops.lstm_rowpar_ok(1024, 512) and ops.lstm_rowpar_ok(1024, 256)
    big = 32 * (cus // 16) * 4                      # row tiles of U = 512: more than 3 per workgroup of the backward (one workgroup per CU)
    assert not ops.lstm_rowpar_ok(big, 512)
    assert not ops.lstm_rowpar_ok(1000, 512)        # not a multiple of the 32-row tile
    assert not ops.lstm_rowpar_ok(1024, 96)         # unit widths: 128 / 256 / 512
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="bf16", seed=23)
    gen._materialize(D)
    gen._ensure_packed()
    assert gen._stack._rowpar(1024, 8) and not gen._stack._rowpar(big, 8) and not gen._stack._rowpar(1024, 8, state0=[None])
    # the C entry point itself refuses the shape (no launch): a fabricated descriptor never reaches a kernel
    T, B, u = 2, 32 * (cus // 16) * 5, 512          # five row tiles per workgroup: more waves than a workgroup of either direction holds
    xproj = torch.empty((T, B, 4 * u), device=DEV)
    h = torch.empty((T, B, u), device=DEV, dtype=torch.bfloat16)
    d = ops.lstm2_fwd_layer(xproj, gen._stack.packed[0]["wh_t"], None, None, None, torch.empty((T, B, u), device=DEV), h, None)
    from multinn_amd import _lib
    ws = torch.zeros(_lib.load().mnn_lstm_rowpar_workspace_bytes(T, B, u), device=DEV, dtype=torch.uint8)
    with pytest.raises(MnnError):
        ops.lstm_rowpar_fwd(T, B, d, 1.0, ws)


@pytest.mark.parametrize("precision", ["fp16", "bf16"])
@pytest.mark.parametrize("layout", ["kblock", "plain"])
def test_rowpar_input_gradient_and_weight_gradient_layouts(precision, layout, monkeypatch):
    """The row-parallel backward with `need_dx` (feedback modes: d loss / d inputs) and the two ways its dz reaches the weight-gradient GEMM --
    K-blocked dz^T (default), plain dz^T -- against the launch-per-step kernels on the same weights:
    every gradient and the input gradient agree to the summation order of the split-K atomics."""
    from multinn_amd import RnnNade
    from multinn_amd.generators import LstmStack
    monkeypatch.setattr(LstmStack, "rowpar_min_batch", 32)
    monkeypatch.setattr(LstmStack, "kblock_wgrads", layout == "kblock")
    B, T = 64, 8
    x = dev(synth(B, T, 41, 0.05))
    a = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
    a._materialize(D)
    a.need_dx = True
    a.build_pianoroll(x, None, True, "train")
    assert a._stack._rowpar(B, T)
    a.backward()
    a._stack.check()
    ga, dxa = a.store.grad.clone(), a._dx.clone()
    b = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23)
    b._materialize(D)
    b.store.theta.copy_(a.store.theta)
    b._stack.rowpar = False
    b._stack.persistent = False
    b.need_dx = True
    b.build_pianoroll(x, None, True, "train")
    b.backward()
    tol = 2e-3 if precision == "fp16" else 2e-2
    assert float((ga - b.store.grad).abs().max()) <= tol * float(b.store.grad.abs().max())
    assert float((dxa - b._dx).abs().max()) <= tol * float(b._dx.abs().max())
    assert float(torch.nn.functional.cosine_similarity(dxa.view(-1), b._dx.view(-1), dim=0)) > 0.9999

