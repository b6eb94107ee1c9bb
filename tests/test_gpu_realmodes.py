"""BASELINE configs[2] and [3] at the reference's REAL layer widths through the mode classes (multinn_amd/modes.py), against the float64 oracle:

    C3  jamming   5 x RnnRBM(88 visibles, 256 hidden, LSTM [512, 256], CD-10), mean track loss, ONE clip over all generators
                  (multinn_jamming.py:40-68,186-245)
    C4  composer  5 x DBNEncoder [88 -> 168 -> 84] -> stacked codes [B, T+1, 420] -> RnnMultiNADE(84, 256, [512, 256]) -> decode
                  (multinn_composer.py:49-87)

fp32: 1e-4 relative on loss / free energy / per-row NLL / every gradient.  fp16 (the benchmarked mode): the same 1e-4 on the forward
quantities BASELINE.json names (free energy, loss, NLL), gradients bounded at 3e-3 (11-bit operands).  bf16: printed, loose bounds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_gpu_modes as TM   # noqa: E402
from oracle import generators as G, rbm as orbm, philox, det   # noqa: E402

FWD_TOL = {"fp32": 1e-4, "fp16": 1e-4, "bf16": 5e-3}
GRAD_TOL = {"fp32": 1e-4, "fp16": 3e-3, "bf16": 5e-2}
P, M, HN, UNITS = 88, 5, 256, [512, 256]


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_c3_jamming_real_widths(precision):
    from multinn_amd import MultINN, AdamOptimizer
    B, T, k = 16, 8, 10
    x = TM.batch(B, T, P, M, 8, rho=0.05)
    m = MultINN(TM.config(P, TM.TRACKS5), TM.params("jamming", gen="RBM", Hn=HN, units=UNITS), mode="jamming", precision=precision)
    m.build(TM.dev(x), lengths=None, is_train=True, mode="train")
    ps = [G.init_rnn_rbm(50 + i, P, P, HN, UNITS, np.float64) for i in range(M)]
    for i, g in enumerate(m.generators):
        ps[i]['bh'] += 0.05 * i
        ps[i]['bv'] += np.log(0.05 / 0.95)
        TM.load_rbm_params(g, ps[i])
    m.build(TM.dev(x), lengths=None, is_train=True, mode="train")
    if precision != "fp32":
        assert all(g._stack._persist(B, T) for g in m.generators), "the persistent recurrence must be the form under test"
    tracks = G.per_track_inputs(x)
    rows = np.array([t * 65536 + b for b in range(B) for t in range(T)])
    grads, losses, worst = [], [], {"free_energy": 0.0, "loss": 0.0, "agree": 1.0}
    for i, g in enumerate(m.generators):
        inp, tgt = tracks[i][:, :-1].astype(np.float64), tracks[i][:, 1:].astype(np.float64)
        du = G.dropout_uniforms(g.seed, B, T, UNITS)
        fw = G.rnn_rbm_forward(inp, tgt, None, ps[i], k, seed=g.seed, keep_prob=0.9, drop_u=du, row_ids=rows)
        vs = g._outputs.cpu().numpy()
        worst["agree"] = min(worst["agree"], float((vs == fw['v_sample']).all(1).mean()))
        fw['v_sample'] = vs.astype(np.float64)              # the device's own chain ends: costs and gradients on identical samples
        cost, F = orbm.free_energy_cost(fw['tgt'], fw['v_sample'], ps[i]['W'], fw['bh_t'], fw['bv_t'])
        worst["free_energy"] = max(worst["free_energy"], TM.rel(g.free_energy.cpu().numpy(), F))
        worst["loss"] = max(worst["loss"], abs(float(g.metrics["batch/loss"]) - cost.mean()) / max(1.0, abs(cost.mean())))
        losses.append(cost.mean())
        grads.append(G.rnn_rbm_backward(fw, ps[i]))
    _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    m.check()
    gerr = 0.0
    for i, g in enumerate(m.generators):
        for name, ref in zip(g.store.names(), TM.rbm_grad_list(grads[i])):
            gerr = max(gerr, TM.rel(g.store.gviews[name].cpu().numpy().reshape(ref.shape), ref / M))
    print(f"\n[C3 {precision}] free energy {worst['free_energy']:.2e}  loss {worst['loss']:.2e}  rows whose Gibbs chain end equals the float64 chain's "
          f"{worst['agree']:.3f}  gradients {gerr:.2e}")
    # the chain ends themselves: the device's CD-10 chains (f32 products, deterministic sigmoid) against the float64 chains driven by the same
    # Philox uniforms -- a draw differs only where a probability sits within f32 rounding of its uniform (measured: every row equal in all
    # three modes; the floor leaves room for one such tie in a thousand rows)
    assert worst["agree"] >= 0.99, worst
    assert worst["free_energy"] < FWD_TOL[precision] and worst["loss"] < FWD_TOL[precision]
    assert abs(float(metrics["batch/loss"]) - np.mean(losses)) < FWD_TOL[precision] * max(1.0, abs(np.mean(losses)))
    assert gerr < GRAD_TOL[precision]


def test_c3_jamming_generators_in_lockstep_equal_one_after_the_other():
    """The jamming mode's five LSTM-RBM generators built and trained in LOCKSTEP (generators.drive_group: one launch of the cluster / CU-resident
    recurrence per layer and direction for all tracks; B = 256 at the reference's widths) against the same model running one generator after
    the other on the two-layer persistent form: the same Gibbs chains (they depend on the Dense outputs only through probabilities: bit-equal
    draws are not required, equal losses to the kernels' 16-bit tolerance are), the same free energies and losses, gradients within the fp16
    bound, and the same weights after two optimiser steps."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T = 256, 6
    x = TM.dev(TM.batch(B, T, P, M, 9, rho=0.05))
    res = []
    for grouped in (True, False):
        m = MultINN(TM.config(P, TM.TRACKS5), TM.params("jamming", gen="RBM", Hn=HN, units=UNITS), mode="jamming", precision="fp16", seed=23)
        m.group_generators = grouped
        opt = AdamOptimizer(0.01)
        m.build(x, lengths=None, is_train=True, mode="train")
        assert bool(getattr(m, "_built_grouped", False)) == grouped
        if grouped:
            assert all(g._ctx["lstm"][0].get("rowpar") for g in m.generators)
        fe = [g.free_energy.clone() for g in m.generators]
        m.train_generators(opt, 0.01)
        m.check()
        grads = [g.store.grad.clone() for g in m.generators]
        loss1 = float(m.generator_loss())
        loss2 = float(m.train_step(x, None, opt))
        m.check()
        res.append((fe, grads, loss1, loss2, [g.store.theta.clone() for g in m.generators]))
    (fa, ga, l1a, l2a, ta), (fb, gb, l1b, l2b, tb) = res
    # (the two recurrence forms agree to 16-bit rounding, so a few Gibbs draws whose uniform sits within that of its probability differ and the
    # chains part there: the free energy of the TARGETS is the tight check, losses and gradients -- which see the chain ends -- the loose ones)
    print(f"\n[C3 lockstep vs sequential] loss {l1a:.5f} / {l1b:.5f}, after a step {l2a:.5f} / {l2b:.5f}")
    for a, b in zip(fa, fb):
        assert float((a - b).abs().max()) < 2e-3 * float(b.abs().max())
    assert abs(l1a - l1b) < 2e-2 * max(1.0, abs(l1b)) and abs(l2a - l2b) < 2e-2 * max(1.0, abs(l2b)), (l1a, l1b, l2a, l2b)
    for a, b in zip(ga, gb):
        assert bool(torch.isfinite(a).all()) and float(torch.nn.functional.cosine_similarity(a, b, dim=0)) > 0.98
    # ragged windows go through the same lockstep path (the recurrences step every row; the row weights carry the lengths)
    ln = torch.from_numpy(np.random.default_rng(3).integers(1, T + 1, B).astype(np.int32)).to(x.device)
    fr = []
    for grouped in (True, False):
        m = MultINN(TM.config(P, TM.TRACKS5), TM.params("jamming", gen="RBM", Hn=HN, units=UNITS), mode="jamming", precision="fp16", seed=23)
        m.group_generators = grouped
        m.build(x, lengths=ln, is_train=True, mode="train")
        assert bool(getattr(m, "_built_grouped", False)) == grouped
        fr.append([g.free_energy.clone() for g in m.generators])
        m.train_generators(AdamOptimizer(0.01), 0.01)
        m.check()
        assert all(bool(torch.isfinite(g.store.grad).all()) for g in m.generators)
    for a, b in zip(*fr):
        assert a.shape == (int(ln.sum()),) and float((a - b).abs().max()) < 2e-3 * float(b.abs().max())


@pytest.mark.parametrize("precision", ["fp32", "fp16", "bf16"])
def test_c4_composer_real_widths(precision):
    from multinn_amd import MultINN, AdamOptimizer
    B, T, E = 32, 8, 84
    x = TM.batch(B, T, P, M, 4, rho=0.1)
    m = MultINN(TM.config(P, TM.TRACKS5), TM.params("composer", enc="DBN", enc_hidden=[168, E], Hn=HN, units=UNITS), mode="composer",
                precision=precision)
    gen = m.generators[0]
    assert type(gen).__name__ == "RnnMultiNADE" and gen.num_dims == E and len(m.encoders) == 5
    m.build(TM.dev(x), lengths=None, is_train=True, mode="train")
    p = G.init_rnn_nade(7, E * M, E, HN, UNITS, np.float64, tracks=M)
    TM.load_nade_params(gen, p)
    m.build(TM.dev(x), lengths=None, is_train=True, mode="train")
    if precision != "fp32":
        assert gen._stack._persist(B, T), "the persistent recurrence must be the form under test"
    tracks = G.per_track_inputs(x)
    N1 = B * (T + 1)
    codes = []
    for i, e in enumerate(m.encoders):                                   # codes: bit-exact against the deterministic checker (f32 kernels in every mode)
        h = tracks[i].reshape(N1, P)
        for l, r in enumerate(e.dbn.rbms):
            ph = det.rbm_hidden(h, r.W.cpu().numpy(), r.bh.cpu().numpy())
            h = (philox.uniform_block(e.seed, philox.STREAM_DBN_ENC, np.arange(N1), (e._sub << 4) | l, ph.shape[1]) < ph).astype(np.uint8)
        assert np.array_equal(e.encodings[-1].cpu().numpy().reshape(N1, E), h), i
        codes.append(h.reshape(B, T + 1, E))
    stack = np.stack(codes, axis=3).reshape(B, T + 1, E * M)
    assert np.array_equal(m._x_encoded_stack.cpu().numpy(), stack)
    inp, tgt = stack[:, :-1].astype(np.float64), stack[:, 1:].astype(np.float64)
    fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, G.dropout_uniforms(gen.seed, B, T, UNITS), tracks=M)
    g = G.rnn_nade_backward(fw, p, tracks=M)
    e_loss = abs(float(m.generator_loss()) - fw['loss']) / fw['loss']
    e_nll = max(TM.rel(gen.log_probs[t].cpu().numpy(), fw['nll'][t]) for t in range(M))
    e_cp = max(float(np.abs(gen.cond_probs[t].cpu().numpy() - fw['cond_p'][t]).max()) for t in range(M))
    m.train_generators(AdamOptimizer(0.01), 0.01)
    m.check()
    ref_g = []
    for W, b in g['lstm']:
        ref_g += [W, b]
    ref_g += [np.stack(g['w_enc']), np.stack(g['w_dec']), g['fc_k'], g['fc_b']]
    gerr = max(TM.rel(gen.store.gviews[name].cpu().numpy().reshape(ref.shape), ref) for name, ref in zip(gen.store.names(), ref_g))
    print(f"\n[C4 {precision}] loss {e_loss:.2e}  per-row NLL {e_nll:.2e}  conditionals (abs) {e_cp:.2e}  gradients {gerr:.2e}")
    assert e_loss < FWD_TOL[precision] and e_nll < FWD_TOL[precision] and e_cp < (1e-4 if precision != "bf16" else 2e-2)
    assert gerr < GRAD_TOL[precision]
    out = m.generate(3)
    assert out.shape == (B, 3, P, M) and out.dtype == torch.uint8


def test_strong_scaling_batch_128_vs_oracle():
    """The strong-scaling leg of bench.py at 8 GPUs runs B = 1024 / 8 = 128 sequences per rank: the two-layer persistent recurrence (not the
    row-parallel form, which needs B >= 512) at FOUR row tiles per launch.  That shape against the float64 oracle at real widths, fp16."""
    import test_gpu_realdims as R
    from multinn_amd import RnnNade
    B, T = 128, 6
    x, p, fw, g = R.oracle(B, T, 0.03)
    gen = RnnNade(R.D, R.HN, R.UNITS, keep_prob=0.9, precision="fp16", seed=23)
    gen._materialize(R.D)
    R.load(gen, p)
    gen.build_pianoroll(R.dev(x), None, is_train=True, mode="train")
    assert gen._stack._persist(B, T) and not gen._stack._rowpar(B, T)
    loss = float(gen.metrics["batch/loss"])
    nll = gen.log_probs.cpu().numpy()
    cp_err = np.abs(gen.cond_probs.cpu().numpy() - fw['cond_p'][0]).max()
    gen.backward()
    gen._stack.check()
    errs = {"loss": abs(loss - fw['loss']) / abs(fw['loss']), "nll": R.rel(nll, fw['nll'][0])}
    for name, ref in zip(gen.store.names(), R.oracle_grads(g)):
        errs[name] = R.rel(gen.store.gviews[name].cpu().numpy().reshape(ref.shape), ref)
    print("\n[B=128 fp16]", {k: f"{v:.2e}" for k, v in errs.items()}, f"cond_p abs {cp_err:.2e}")
    assert errs["loss"] < 1e-4 and errs["nll"] < 1e-4 and cp_err < 1e-4
    assert all(v < 3e-3 for v in errs.values()), errs
