"""BASELINE.json configurations run AS CONFIGURATIONS through the mode classes (multinn_amd/modes.py) against the CPU oracle:

    C1  joint     PassEncoder + LSTM-RBM, 1 track
    C2  joint     PassEncoder + LSTM-NADE (the fused piano-roll path) + the joint `/ num_tracks` metrics
    C3  jamming   5 per-track LSTM-RBM generators, CD-10, mean track loss, ONE global-norm clip over all of them
    C4  composer  5 DBNEncoders -> stacked codes -> RnnMultiNADE -> decode
    C5  feedback-rnn sampling scan (per-track generators + recurrent feedback), incl. T = 512 generated steps

fp32 mode: 1e-4 relative on losses / free energies / gradients (BASELINE.json gate); Bernoulli draws bit-exact against the
deterministic C checker where the probabilities are computed by the deterministic kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import generators as G, nade as onade, rbm as orbm, lstm as olstm, philox, det   # noqa: E402
from oracle import tf_semantics as S   # noqa: E402

DEV = "cuda:0"
TRACKS5 = ["Drums", "Piano", "Guitar", "Bass", "Strings"]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(1e-30, np.abs(b).max())


def config(P, tracks, num_pixels=1, beat_resolution=4):
    return {"model_name": "t", "data": {"pitch_range": {"lowest": 0, "highest": P // num_pixels}, "instruments": list(tracks),
                                         "beat_resolution": beat_resolution},
            "training": {"num_pixels": num_pixels, "random_seed": 23}}


def params(mode, enc="Pass", enc_hidden=None, gen="NADE", Hn=16, units=(32, 32), feedback=None, keep_prob=0.9):
    return {"mode": mode, "tune_encoder": False, "keep_prob": keep_prob, "encoder": {"type": enc, "num_hidden": enc_hidden},
            "generator": {"type": gen, "num_hidden": Hn, "num_hidden_rnn": list(units), "feedback": feedback}}


def batch(B, T, P, M, seed, rho=0.25):
    return (np.random.default_rng(seed).random((B, T, P, M)) < rho).astype(np.uint8)


def load_rbm_params(gen, p):
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W.astype(np.float32))); s[f"rnn/cell_{l}/bias"].copy_(dev(b.astype(np.float32)))
    for kk in ("W", "bh", "bv"):
        s[f"rbm/{kk}"].copy_(dev(p[kk].astype(np.float32)))
    s["Wuh"].copy_(dev(p['Wuh'].astype(np.float32))); s["Wuv"].copy_(dev(p['Wuv'].astype(np.float32)))
    gen._packed_step = -1


def load_nade_params(gen, p):
    s = gen.store
    for l, (W, b) in enumerate(p['lstm']):
        s[f"rnn/cell_{l}/kernel"].copy_(dev(W.astype(np.float32))); s[f"rnn/cell_{l}/bias"].copy_(dev(b.astype(np.float32)))
    s["nade/w_enc"].copy_(dev(np.stack(p['w_enc']).astype(np.float32)))
    s["nade/w_dec"].copy_(dev(np.stack(p['w_dec']).astype(np.float32)))
    s["dense/kernel"].copy_(dev(p['fc_k'].astype(np.float32)))
    s["dense/bias"].copy_(dev(p['fc_b'].astype(np.float32)))
    gen._packed_step = -1


def rbm_grad_list(g):
    """Order of RnnRBM's variables: rbm [W, bv, bh], rnn, Wuh, Wuv (rnn_rbm.py:135-138)."""
    out = [g['W'], g['bv'], g['bh']]
    for W, b in g['lstm']:
        out += [W, b]
    return out + [g['Wuh'], g['Wuv']]


def rbm_param_list(p):
    out = [p['W'], p['bv'], p['bh']]
    for W, b in p['lstm']:
        out += [W, b]
    return out + [p['Wuh'], p['Wuv']]


def log_loss_rows(t, p):
    return S.log_loss(t.astype(np.float64), p.astype(np.float64)).sum(1)


# ------------------------------------------------------------------------------------------------------------------------
def test_c1_joint_pass_lstm_rbm_one_track():
    """BASELINE configs[0]: Joint PassEncoder + LSTM-RBM, 1 track, batch 16, T 64 (feature width cut to keep the float64 oracle quick)."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P, M, Hn, units, k = 16, 64, 20, 1, 24, [32, 32], 10
    x = batch(B, T, P, M, 3, rho=0.15)
    m = MultINN(config(P, ["Piano"]), params("joint", gen="RBM", Hn=Hn, units=units), mode="joint", precision="fp32")
    gen = m.generators[0]
    assert type(gen).__name__ == "RnnRBM" and gen.k == 10 and gen.num_dims == P
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    p = G.init_rnn_rbm(5, P, P, Hn, units, np.float64)
    p['bh'] += 0.1; p['bv'] -= 0.2
    load_rbm_params(gen, p)
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    inp, tgt = G.joint_inputs(x)
    rows = np.array([t * 65536 + b for b in range(B) for t in range(T)])
    du = G.dropout_uniforms(gen.seed, B, T, units)
    fw = G.rnn_rbm_forward(inp.astype(np.float64), tgt.astype(np.float64), None, p, k, seed=gen.seed, keep_prob=0.9, drop_u=du, row_ids=rows)
    vs = gen._outputs.cpu().numpy()
    agree = (vs == fw['v_sample']).all(1).mean()
    assert agree >= 0.9, agree
    fw['v_sample'] = vs.astype(np.float64)                       # the device's own chain end: a rare u ~ p flip must not hide arithmetic errors
    cost, F = orbm.free_energy_cost(fw['tgt'], fw['v_sample'], p['W'], fw['bh_t'], fw['bv_t'])
    assert rel(gen.free_energy.cpu().numpy(), F) < 1e-4 and rel(gen.cost.cpu().numpy(), cost) < 1e-4
    assert abs(float(m.generator_loss()) - cost.mean()) < 1e-4 * max(1.0, abs(cost.mean()))
    pv = gen.cond_probs.cpu().numpy()
    assert rel(gen.reconstruction_cost.cpu().numpy(), log_loss_rows(fw['tgt'], pv)) < 1e-5          # mnn_log_loss_rows (rbm.py:124-129)
    # encoder-level ("global") metrics of the joint mode: log-loss of the DECODED hard outputs, / num_tracks (multinn_joint.py:177-184)
    glob = m.metrics
    ref = log_loss_rows(fw['tgt'], vs).mean() / M
    assert abs(float(glob["batch/loss"]) - ref) < 1e-5 * ref and abs(float(glob["log_likelihood"]) - ref) < 1e-5 * ref
    # backward + clip + TF-Adam through the mode's train_generators
    g = G.rnn_rbm_backward(fw, p)
    _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    assert "global" in metrics and float(metrics["global"]["batch/loss"]) == float(glob["batch/loss"])
    opt = G.new_opt(rbm_param_list(p))
    gn = G.apply_clip_adam(rbm_param_list(p), rbm_grad_list(g), opt, lr=0.01)
    assert abs(float(gen._grad_sumsq.sqrt()) - gn) < 1e-4 * gn
    for name, ref_p in zip(gen.store.names(), rbm_param_list(p)):
        assert np.abs(gen.store[name].cpu().numpy().reshape(ref_p.shape) - ref_p).max() < 2e-4, name
    out = m.generate(3)
    assert out.shape == (B, 3, P, M) and out.dtype == torch.uint8


def test_c2_joint_nade_global_metrics_divide_by_tracks():
    """multinn_joint.py:159-186: the joint mode's encoder-level loss / NLL / perplexity are divided by the number of tracks; the
    generator's own loss (statistical.py:34) is not."""
    from multinn_amd import MultINN
    B, T, P, M, Hn, units = 6, 5, 4, 3, 20, [32, 64]
    x = batch(B, T, P, M, 1)
    lengths = np.array([5, 2, 4, 5, 1, 3], np.int32)
    m = MultINN(config(P, TRACKS5[:M]), params("joint", Hn=Hn, units=units), mode="joint", precision="fp32")
    m.build(dev(x), lengths=dev(lengths), is_train=False, mode="eval")
    gen = m.generators[0]
    p = G.init_rnn_nade(3, P * M, P * M, Hn, units, np.float64)
    load_nade_params(gen, p)
    m.build(dev(x), lengths=dev(lengths), is_train=False, mode="eval")
    inp, tgt = G.joint_inputs(x.astype(np.float64))
    fw = G.rnn_nade_forward(inp, tgt, lengths, p, 1.0, None)
    assert abs(float(m.generator_loss()) - fw['loss']) < 1e-4 * fw['loss']
    hard = (fw['cond_p'][0] >= 0.5).astype(np.float64)
    tflat = S.flatten_maybe_padded_sequences(tgt, lengths)
    rows = log_loss_rows(tflat, hard)
    glob = m.metrics
    assert abs(float(glob["batch/loss"]) - rows.mean() / M) < 1e-5 * rows.mean()
    assert abs(float(glob["log_likelihood"]) - rows.mean() / M) < 1e-5 * rows.mean()
    assert abs(glob["accuracy"] - (hard == tflat).mean()) < 1e-6
    assert float(m.loss) == float(glob["batch/loss"])
    out = m.generate(2)
    assert out.shape == (B, 2, P, M)
    # the generic path (explicit encoder build / encode / slicing) gives the same numbers as the fused piano-roll kernel
    m._fused = lambda: False
    m.build(dev(x), lengths=dev(lengths), is_train=False, mode="eval")
    assert abs(float(m.generator_loss()) - fw['loss']) < 1e-4 * fw['loss']
    assert abs(float(m.metrics["batch/loss"]) - rows.mean() / M) < 1e-5 * rows.mean()


def test_c3_jamming_five_lstm_rbm_cd10():
    """BASELINE configs[2]: per-track LSTM-RBM generators (CD-10 Gibbs), 5 tracks; the optimised loss is the MEAN track loss with one
    clip_by_global_norm over ALL generators' variables (multinn_jamming.py:186-245)."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P, M, Hn, units, k = 4, 5, 12, 5, 16, [32, 32], 10
    x = batch(B, T, P, M, 8)
    m = MultINN(config(P, TRACKS5), params("jamming", gen="RBM", Hn=Hn, units=units), mode="jamming", precision="fp32")
    assert [type(g).__name__ for g in m.generators] == ["RnnRBM"] * 5 and len({g.seed for g in m.generators}) == 5
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    ps = [G.init_rnn_rbm(50 + i, P, P, Hn, units, np.float64) for i in range(M)]
    for i, g in enumerate(m.generators):
        ps[i]['bh'] += 0.05 * i
        load_rbm_params(g, ps[i])
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    tracks = G.per_track_inputs(x)                                       # multi_encoder_nn.py:66-76
    rows = np.array([t * 65536 + b for b in range(B) for t in range(T)])
    fws, grads, losses = [], [], []
    for i, g in enumerate(m.generators):
        assert g.grad_scale == 1.0 / M
        inp, tgt = tracks[i][:, :-1].astype(np.float64), tracks[i][:, 1:].astype(np.float64)       # multinn_jamming.py:61-65
        du = G.dropout_uniforms(g.seed, B, T, units)
        fw = G.rnn_rbm_forward(inp, tgt, None, ps[i], k, seed=g.seed, keep_prob=0.9, drop_u=du, row_ids=rows)
        vs = g._outputs.cpu().numpy()
        assert (vs == fw['v_sample']).all(1).mean() >= 0.85
        fw['v_sample'] = vs.astype(np.float64)
        cost, F = orbm.free_energy_cost(fw['tgt'], fw['v_sample'], ps[i]['W'], fw['bh_t'], fw['bv_t'])
        assert rel(g.free_energy.cpu().numpy(), F) < 1e-4
        assert abs(float(g.metrics["batch/loss"]) - cost.mean()) < 1e-4 * max(1.0, abs(cost.mean()))
        losses.append(cost.mean())
        fws.append(fw)
        grads.append(G.rnn_rbm_backward(fw, ps[i]))
    assert abs(float(m.generator_loss()) - np.mean(losses)) < 1e-4 * max(1.0, abs(np.mean(losses)))
    _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    assert abs(float(metrics["batch/loss"]) - np.mean(losses)) < 1e-4 * max(1.0, abs(np.mean(losses)))
    # d(mean_i L_i)/d theta_i = g_i / M; ONE global norm over all five generators
    all_p = [a for p in ps for a in rbm_param_list(p)]
    all_g = [a / M for g in grads for a in rbm_grad_list(g)]
    for i, g in enumerate(m.generators):
        for name, ref in zip(g.store.names(), rbm_grad_list(grads[i])):
            assert rel(g.store.gviews[name].cpu().numpy().reshape(ref.shape), ref / M) < 1e-4, (i, name)
    gn = G.apply_clip_adam(all_p, all_g, G.new_opt(all_p), lr=0.01)
    assert abs(float(m._grad_sumsq.sqrt()) - gn) < 1e-4 * gn
    for i, g in enumerate(m.generators):
        for name, ref in zip(g.store.names(), rbm_param_list(ps[i])):
            assert np.abs(g.store[name].cpu().numpy().reshape(ref.shape) - ref).max() < 2e-4, (i, name)
    glob = metrics["global"]
    assert {"batch/loss", "log_likelihood", "accuracy"} <= set(glob.keys())
    out = m.generate(3)
    assert out.shape == (B, 3, P, M) and out.dtype == torch.uint8
    # `separate_losses` passed to train_generators is honoured even when it differs from the attribute the batch was built with
    # (multinn_jamming.py:213-221: run_optimizer = separate_losses): every generator then steps on ITS OWN loss, gradient g_i instead of g_i / M
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    opt0 = AdamOptimizer(0.0)                                            # zero step size: the weights stay put between the two calls
    mean_grads = []
    for g in m.generators:                                               # built for the mean track loss: d (L_i / M)
        g.backward()
        mean_grads.append(g.store.grad.clone())
    m.train_generators(opt0, 0.0, separate_losses=True)
    for g, r in zip(m.generators, mean_grads):
        assert float((g.store.grad - M * r).abs().max()) <= 1e-4 * float((M * r).abs().max())
    # pre-training without separate losses = init ops + the joint step on the mean track loss (multinn_jamming.py:235-241 runs for both)
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    steps = [g.store.step for g in m.generators]
    init_ops, update_ops, _, _, _ = m.pretrain_generators(opt0, 0.0)
    assert len(init_ops) == M and update_ops == [] and [g.store.step for g in m.generators] == [s_ + 1 for s_ in steps]


def test_c4_composer_dbn_encoders_multinade():
    """BASELINE configs[3]: per-track DBNEncoders -> stack on axis 3 -> [B,T+1,E*M] -> [:, :-1] / [:, 1:] -> RnnMultiNADE -> decode
    (multinn_composer.py:73-87,114-151, multi_encoder_nn.py:66-115).  Codes are checked bit-exact against the deterministic checker,
    the generator against the float64 oracle on those codes."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P, M, Hn, units, E = 5, 4, 10, 5, 12, [32, 32], 6
    x = batch(B, T, P, M, 4, rho=0.3)
    m = MultINN(config(P, TRACKS5), params("composer", enc="DBN", enc_hidden=[8, E], Hn=Hn, units=units), mode="composer", precision="fp32")
    gen = m.generators[0]
    assert type(gen).__name__ == "RnnMultiNADE" and gen.num_dims == E and len(m.encoders) == 5
    assert all(type(e).__name__ == "DBNEncoder" and e.dbn.rbms[0].k == 2 for e in m.encoders)
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    p = G.init_rnn_nade(7, E * M, E, Hn, units, np.float64, tracks=M)
    load_nade_params(gen, p)
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    # encoders: sampled binary codes of the zero-padded per-track sequences (the padded step is encoded too)
    tracks = G.per_track_inputs(x)
    N1 = B * (T + 1)
    codes = []
    for i, e in enumerate(m.encoders):
        h = tracks[i].reshape(N1, P)
        for l, r in enumerate(e.dbn.rbms):
            ph = det.rbm_hidden(h, r.W.cpu().numpy(), r.bh.cpu().numpy())
            h = (philox.uniform_block(e.seed, philox.STREAM_DBN_ENC, np.arange(N1), (e._sub << 4) | l, ph.shape[1]) < ph).astype(np.uint8)
        assert np.array_equal(e.encodings[-1].cpu().numpy().reshape(N1, E), h), i
        codes.append(h.reshape(B, T + 1, E))
    stack = np.stack(codes, axis=3).reshape(B, T + 1, E * M)             # multinn_composer.py:73-80: feature e*M+m
    assert np.array_equal(m._x_encoded_stack.cpu().numpy(), stack)
    inp, tgt = stack[:, :-1].astype(np.float64), stack[:, 1:].astype(np.float64)
    du = G.dropout_uniforms(gen.seed, B, T, units)
    fw = G.rnn_nade_forward(inp, tgt, None, p, 0.9, du, tracks=M)
    g = G.rnn_nade_backward(fw, p, tracks=M)
    assert abs(float(m.generator_loss()) - fw['loss']) < 1e-4 * fw['loss']
    for t in range(M):
        assert rel(gen.log_probs[t].cpu().numpy(), fw['nll'][t]) < 1e-4
    # decode: hard generator outputs through each track's DBN, reconstruction cost against the raw targets
    glob = m.metrics
    hard = [(c >= 0.5).astype(np.uint8) for c in fw['cond_p']]
    for t in range(M):
        assert np.array_equal(m._x_hidden[t].cpu().numpy().astype(np.uint8), hard[t]), t
        assert m._outputs_probs[t].shape == (B * T, P) and m._outputs[t].shape == (B * T, P)
    assert {"batch/loss", "log_likelihood", "free_energy", "accuracy"} <= set(glob.keys())
    _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    ref_g = []
    for W, b in g['lstm']:
        ref_g += [W, b]
    ref_g += [np.stack(g['w_enc']), np.stack(g['w_dec']), g['fc_k'], g['fc_b']]
    for name, ref in zip(gen.store.names(), ref_g):
        assert rel(gen.store.gviews[name].cpu().numpy().reshape(ref.shape), ref) < 1e-4, name
    assert float(metrics["global"]["batch/loss"]) == float(glob["batch/loss"])
    out = m.generate(3)
    assert out.shape == (B, 3, P, M) and out.dtype == torch.uint8 and torch.equal(out, m.generate(3))
    # the encoders train through the mode too (multi_encoder_nn.py:155-195)
    _, _, em, _, _ = m.train_encoders(None, 0.05, layer=1)
    assert "batch/loss" in em and np.isfinite(float(em["batch/loss"]))


def _feedback_model(P, M, Hn, units, fb_units, precision, B, Ti, seed=14):
    from multinn_amd import MultINN
    x = batch(B, Ti, P, M, seed, rho=0.3)
    m = MultINN(config(P, TRACKS5[:M]), params("feedback-rnn", Hn=Hn, units=units, feedback=fb_units, keep_prob=0.9), mode="feedback-rnn",
                precision=precision)
    m.build(dev(x), lengths=None, is_train=False, mode="generate")
    gparams = []
    for i, g in enumerate(m.generators):
        p = G.init_rnn_nade(60 + i, P + fb_units[-1], P, Hn, units, np.float64)
        p['fc_b'][Hn:] = np.log(0.15 / 0.85)                        # piano-roll-like conditionals instead of coin flips
        load_nade_params(g, p)
        gparams.append(p)
    fb = m._feedback_layer
    fb_layers = [(fb.store[f"feedback/rnn/cell_{l}/kernel"].cpu().numpy().astype(np.float64),
                  fb.store[f"feedback/rnn/cell_{l}/bias"].cpu().numpy().astype(np.float64)) for l in range(len(fb_units))]
    return m, x, gparams, fb_layers, [g.seed for g in m.generators]


def test_c5_feedback_rnn_mode_sampling_scan():
    """BASELINE configs[4] through the mode class: intro pass, then per step M x NADE sample -> stack -> feedback LSTM step ->
    M x {LSTM step on concat(sample_i, feedback) -> Dense} (multinn_feedback.py:120-218, multinn_feedback_rnn.py:41-79)."""
    m, x, gparams, fb_layers, seeds = _feedback_model(8, 3, 16, [32, 32], [64, 32], "fp32", 4, 3)
    steps = 6
    out = m.generate(steps)
    assert out.shape == (4, steps, 8, 3) and out.dtype == torch.uint8
    got = out.cpu().numpy()
    probs, us = G.feedback_rnn_teacher_forced(x, got, gparams, fb_layers, seeds)
    bad = (us < probs) != (got > 0)
    assert not (bad & (np.abs(us - probs) > 2e-5)).any()
    assert np.array_equal(got, det.feedback_rnn_generate(x, steps, gparams, fb_layers, seeds))      # every cell == the deterministic checker's scan
    assert torch.equal(out, m.generate(steps))
    # eval build of the same mode: generator inputs = concat(track code, feedback vector)[:, :-1] (multinn_feedback.py:85-94)
    m.build(dev(x), lengths=None, is_train=False, mode="eval")
    enc = np.concatenate([np.zeros((4, 1, 8, 3)), x.astype(np.float64)], 1)
    x_fb, _, _ = olstm.seq_fwd(enc.reshape(4, 4, 24), fb_layers)
    for i, g in enumerate(m.generators):
        inp = np.concatenate([enc[..., i], x_fb], -1)[:, :-1]
        fw = G.rnn_nade_forward(inp, enc[..., i][:, 1:], None, gparams[i], 1.0, None)
        assert abs(float(g.metrics["batch/loss"]) - fw['loss']) < 1e-4 * fw['loss'], i
    m.build(dev(x), lengths=None, is_train=True, mode="train")          # trainable too: test_feedback_modes_train_step_vs_oracle


def test_c5_feedback_rnn_512_generated_steps():
    """C5 at its real length and widths (default_feedback_rnn.yaml: generators [256,256], NADE 256, feedback LSTM [256,128], P = 88,
    5 tracks): 512 generated steps in ONE captured scan (17.9 k graph nodes), every draw replayed teacher-forced by the float64
    oracle (fp32 mode), and run-to-run reproducible; `sub` = the generated step in every RNG counter."""
    P, M, B, Ti, steps = 88, 5, 4, 8, 512
    m, x, gparams, fb_layers, seeds = _feedback_model(P, M, 256, [256, 256], [256, 128], "fp32", B, Ti, seed=21)
    out = m.generate(steps)
    assert out.shape == (B, steps, P, M)
    got = out.cpu().numpy()
    probs, us = G.feedback_rnn_teacher_forced(x, got, gparams, fb_layers, seeds)
    bad = (us < probs) != (got > 0)
    assert not (bad & (np.abs(us - probs) > 5e-5)).any(), int((bad & (np.abs(us - probs) > 5e-5)).sum())
    ref = det.feedback_rnn_generate(x, steps, gparams, fb_layers, seeds)         # and bit for bit: all 512 x 4 x 88 x 5 cells
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} cells differ; first at {np.argwhere(got != ref)[:3].tolist()}"
    dens = got.mean()
    assert 0.01 < dens < 0.6, dens
    late = got[:, 384:]
    assert late.any() and not np.array_equal(got[:, 384:448], got[:, 448:512])      # still drawing new material at the end
    assert torch.equal(out, m.generate(steps))
    for g in m.generators:
        g._stack.check()


def test_c5_feedback_rnn_512_steps_bf16_reproducible():
    """The benchmarked precision at the same length: two generate() calls give the same 512 x 5 x 88 draws, every step's draws differ
    from the previous step's counters (no stale `sub`), and the persistent-launch status words stay clean."""
    P, M, B, Ti, steps = 88, 5, 8, 8, 512
    m, x, _, _, _ = _feedback_model(P, M, 256, [256, 256], [256, 128], "bf16", B, Ti, seed=22)
    a = m.generate(steps)
    b = m.generate(steps)
    assert torch.equal(a, b) and a.shape == (B, steps, P, M)
    got = a.cpu().numpy()
    assert 0.01 < got.mean() < 0.6
    same = [(got[:, s] == got[:, s - 1]).all() for s in range(1, steps)]
    assert sum(same) < steps // 8
    for g in m.generators:
        g._stack.check()


def test_rbm_visible_bias_init_and_rnn_rbm_pretrain():
    """A17: visible_bias_init_ops (rbm.py:286-297) and RnnRBM.pretrain (rnn_rbm.py:299-322: one CD-k update of the RBM module on
    the flattened inputs) against oracle/rbm.py."""
    from multinn_amd import RnnRBM
    B, T, D, Hn, units, k = 5, 6, 12, 10, [32, 32], 3
    x = batch(B, T, D, 1, 11, rho=0.3)
    inp, tgt = G.joint_inputs(x)
    lengths = np.array([6, 3, 5, 6, 2], np.int32)
    gen = RnnRBM(D, Hn, units, keep_prob=1.0, k=k, precision="fp32", seed=17)
    gen.build(dev(inp), dev(tgt), dev(lengths), True, "train")
    W0, bh0, bv0 = (gen._rbm.W.cpu().numpy().astype(np.float64), gen._rbm.bh.cpu().numpy().astype(np.float64),
                    gen._rbm.bv.cpu().numpy().astype(np.float64))
    flat = S.flatten_maybe_padded_sequences(inp, lengths).astype(np.float64)
    N = flat.shape[0]
    init_ops, update_ops, metrics, metrics_upd, summaries = gen.pretrain(None, 0.1)
    assert update_ops == [] and len(init_ops) == 1 and "batch/loss" in metrics
    rows = np.arange(N)
    u_h, u_v = G.gibbs_uniforms(17, rows, k, Hn, D, 0)
    u0 = philox.uniform_block(17, philox.STREAM_RBM_H, rows, k, Hn)
    uk = philox.uniform_block(17, philox.STREAM_RBM_H, rows, k + 1, Hn)
    dW, dbv, dbh = orbm.cd_update(flat, W0, bh0, bv0, k, 0.1, u_h, u_v, u0, uk)
    gW, gbv, gbh = gen._cd_gradients
    assert rel(gW.cpu().numpy(), dW) < 1e-4 and rel(gbv.cpu().numpy(), dbv) < 1e-4 and rel(gbh.cpu().numpy(), dbh) < 1e-4
    assert rel(gen._rbm.W.cpu().numpy(), W0 + dW) < 1e-5 and rel(gen._rbm.bh.cpu().numpy(), bh0 + dbh) < 1e-5
    assert np.abs(gen._rbm.bv.cpu().numpy() - (bv0 + dbv)).max() < 1e-6
    # the init op is returned unexecuted (the reference runs init_ops once, train_encoders.py:150-153); running it assigns bv
    init_ops[0]()
    assert rel(gen._rbm.bv.cpu().numpy().reshape(-1), orbm.visible_bias_init(flat)) < 1e-5
    pm = flat.mean(0)
    assert np.allclose(gen._rbm.bv.cpu().numpy().reshape(-1), np.log(1e-6 + pm / (1 - pm)), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode", ["feedback", "feedback-rnn"])
def test_feedback_modes_train_step_vs_oracle(mode):
    """Training the feedback modes (multinn_feedback.py:54-101 + multinn_jamming.py:186-245): the mean track loss is minimised over the M
    generators AND the feedback module, whose output is columns [E, E + F) of every generator's input -- so each generator hands back the
    gradient wrt its inputs and the module back-propagates their sum.  fp32, keep_prob = 1: every gradient (generators / M, Dense or LSTM
    feedback module) and the joint global norm against the oracle composed from its LSTM / NADE pieces."""
    from multinn_amd import MultINN, AdamOptimizer
    P, M, Hn, units, B, Ti = 8, 3, 12, [32, 32], 4, 5
    fb_units = [24] if mode == "feedback" else [64, 32]
    F = fb_units[-1]
    x = batch(B, Ti, P, M, 31, rho=0.3)
    m = MultINN(config(P, TRACKS5[:M]), params(mode, Hn=Hn, units=units, feedback=fb_units, keep_prob=1.0), mode=mode, precision="fp32")
    m.build(dev(x), lengths=None, is_train=True, mode="train")
    gparams = []
    for i, g in enumerate(m.generators):
        p = G.init_rnn_nade(80 + i, P + F, P, Hn, units, np.float64)
        load_nade_params(g, p)
        gparams.append(p)
    fb = m._feedback_layer
    f64 = lambda t: t.cpu().numpy().astype(np.float64)
    m.build(dev(x), lengths=None, is_train=True, mode="train")          # with the loaded weights
    enc = np.concatenate([np.zeros((B, 1, P, M)), x.astype(np.float64)], 1)            # PassEncoder codes, zero first step
    stack = enc.reshape(B, Ti + 1, P * M)
    if mode == "feedback":
        Ws = [(f64(fb.store[f"feedback/dnn/dense_{l}/kernel"]), f64(fb.store[f"feedback/dnn/dense_{l}/bias"])) for l in range(len(fb_units))]
        acts = [stack.reshape(-1, P * M)]
        for W, b in Ws:
            acts.append(1.0 / (1.0 + np.exp(-(acts[-1] @ W + b))))
        x_fb = acts[-1].reshape(B, Ti + 1, F)
    else:
        fb_layers = [(f64(fb.store[f"feedback/rnn/cell_{l}/kernel"]), f64(fb.store[f"feedback/rnn/cell_{l}/bias"])) for l in range(len(fb_units))]
        x_fb, _, fcache = olstm.seq_fwd(stack, fb_layers)
    d_fb = np.zeros((B, Ti + 1, F))
    refs, loss = [], 0.0
    for i in range(M):
        inp = np.concatenate([enc[..., i], x_fb], -1)[:, :-1]
        fw = G.rnn_nade_forward(inp, enc[..., i][:, 1:], None, gparams[i], 1.0, None)
        gi = G.rnn_nade_backward(fw, gparams[i])
        loss += fw['loss'] / M
        d_fb[:, :-1] += gi['dx'][..., P:] / M
        refs.append([a / M for a in ([w_ for pair in gi['lstm'] for w_ in pair] + [np.stack(gi['w_enc']), np.stack(gi['w_dec']), gi['fc_k'], gi['fc_b']])])
    if mode == "feedback":
        fb_ref, dy = [], d_fb.reshape(-1, F)
        for l in range(len(Ws) - 1, -1, -1):
            dz = dy * acts[l + 1] * (1 - acts[l + 1])
            fb_ref = [acts[l].T @ dz, dz.sum(0)] + fb_ref
            dy = dz @ Ws[l][0].T
    else:
        _, lg = olstm.seq_bwd(d_fb, fcache)
        fb_ref = [a for pair in lg for a in pair]
    _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    assert abs(float(metrics["batch/loss"]) - loss) < 1e-4 * abs(loss)
    sq = 0.0
    for i, g in enumerate(m.generators):
        for name, r_ in zip(g.store.names(), refs[i]):
            got = g.store.gviews[name].cpu().numpy().reshape(r_.shape)
            assert rel(got, r_) < 1e-4, (i, name)
            sq += float((r_ ** 2).sum())
    for name, r_ in zip(fb.store.names(), fb_ref):
        got = fb.store.gviews[name].cpu().numpy().reshape(r_.shape)
        assert rel(got, r_) < 1e-4, name
        assert np.abs(r_).max() > 0                                                     # the module really receives a gradient
        sq += float((r_ ** 2).sum())
    assert abs(float(m._grad_sumsq.sqrt()) - np.sqrt(sq)) < 1e-4 * np.sqrt(sq)         # ONE global norm over generators + feedback module
    before = float(metrics["batch/loss"])
    for _ in range(5):
        m.build(dev(x), lengths=None, is_train=True, mode="train")
        _, _, metrics, _, _ = m.train_generators(AdamOptimizer(0.01), 0.01)
    assert float(metrics["batch/loss"]) < before


@pytest.mark.parametrize("mode,enc,gen", [("jamming", "DBN", "RBM"), ("feedback-rnn", "Pass", "NADE"), ("feedback", "DBN", "NADE"),
                                          ("composer", "DBN", "NADE"), ("joint", "Pass", "NADE")])
def test_mode_checkpoint_round_trip(mode, enc, gen, tmp_path):
    """model.py:180-234 through a MODE: save after a few optimiser steps, load into a FRESH model before it has seen a batch (the driver's
    resume order: build_model, load, fit), and find every store bit-equal -- parameters, both Adam slots and the step count of every
    per-track generator (they all carry the same default name), of every per-track DBN encoder and of the feedback module, whose variables
    only exist after the first build.  Then the two models take the same next step."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P, M = 4, 5, 12, 5
    x = dev(batch(B, T, P, M, 12))
    kw = dict(enc=enc, enc_hidden=[10, 8] if enc == "DBN" else None, gen=gen, Hn=16, units=[32, 32],
              feedback=[32] if mode.startswith("feedback") else None)
    a = MultINN(config(P, TRACKS5), params(mode, **kw), mode=mode, precision="fp32")
    opt = AdamOptimizer(0.01)
    for _ in range(2):
        a.train_step(x, None, opt)
    path = a.save(None, str(tmp_path), global_step=2)
    assert path.endswith(".pt") and len(list(tmp_path.iterdir())) == 1          # ONE file: no per-model files overwriting one another
    b = MultINN(config(P, TRACKS5), params(mode, **kw), mode=mode, precision="fp32")
    for e in b.encoders:                            # the stores that exist before the first build must really be overwritten by load()
        if getattr(e, "store", None) is not None and e.store.theta is not None:
            e.store.theta.add_(1.0)
    assert all(g.store.theta is None for g in b.generators)          # the generators' variables do not exist yet: load() declares them
    assert b.load(None, str(tmp_path)) is True
    assert MultINN(config(P, TRACKS5), params(mode, **kw), mode=mode, precision="fp32").load(None, str(tmp_path / "nowhere")) is False

    def stores(m):
        out = [g.store for g in m.generators] + [e.store for e in m.encoders if getattr(e, "store", None) is not None and e.store.theta is not None]
        fl = getattr(m, "_feedback_layer", None)
        return out + ([fl.store] if fl is not None else [])
    sa, sb = stores(a), stores(b)
    assert len(sa) == len(sb) and len(sa) >= len(a.generators) + (1 if mode.startswith("feedback") else 0)
    for p_, q_ in zip(sa, sb):
        assert p_.names() == q_.names() and p_.step == q_.step and int(q_.step_dev) == q_.step
        assert torch.equal(p_.theta, q_.theta) and torch.equal(p_.m, q_.m) and torch.equal(p_.v, q_.v)
    la = float(a.train_step(x, None, opt))
    lb = float(b.train_step(x, None, opt))
    assert abs(la - lb) <= 1e-4 * max(1.0, abs(la)), (la, lb)
    b.check()
    with pytest.raises(ValueError):
        other = "joint" if mode != "joint" else "jamming"
        MultINN(config(P, TRACKS5), params(other, gen="NADE", Hn=16, units=[32, 32]), mode=other, precision="fp32").load(None, str(tmp_path))


def test_load_encoders_restores_the_encoder_stores_from_the_mode_checkpoint(tmp_path):
    """The reference flow train_encoders.py `model.save(encoders_dir)` -> train.py:126 `model.load_encoders(encoders_dir)`: a fresh composer
    model restores its per-track DBN encoder variables -- and ONLY those -- from the one checkpoint file `save()` writes; a directory with
    nothing in it returns False."""
    from multinn_amd import MultINN
    B, T, P, M = 4, 5, 12, 5
    x = dev(batch(B, T, P, M, 13))
    kw = dict(enc="DBN", enc_hidden=[10, 8], gen="NADE", Hn=16, units=[32, 32])
    a = MultINN(config(P, TRACKS5), params("composer", **kw), mode="composer", precision="fp32")
    for e in a.encoders:                                # stand-in for the pre-training: distinct, non-initial encoder weights
        e.store.theta.add_(torch.randn_like(e.store.theta) * 0.1)
    a.save(None, str(tmp_path))
    b = MultINN(config(P, TRACKS5), params("composer", **kw), mode="composer", precision="fp32")
    before = [e.store.theta.clone() for e in b.encoders]
    assert b.load_encoders(None, str(tmp_path / "nowhere")) is False
    assert all(torch.equal(e.store.theta, t0) for e, t0 in zip(b.encoders, before))
    assert b.load_encoders(None, str(tmp_path)) is True
    for ea, eb in zip(a.encoders, b.encoders):
        assert torch.equal(ea.store.theta, eb.store.theta)
    assert all(g.store.theta is None for g in b.generators)          # the generators were not touched
    b.train_step(x, None, __import__("multinn_amd").AdamOptimizer(0.01))
    b.check()


def test_check_raises_for_skipped_steps_of_generator_encoder_and_feedback_stores(tmp_path):
    """an optimiser step the device skipped (non-finite gradient norm) must surface through MultINNCore.check() whichever
    store it happened in -- the per-track generators', the feedback module's and the ENCODERS' (which also receive store.skipped)."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P, M = 4, 5, 12, 5
    x = dev(batch(B, T, P, M, 21))
    a = MultINN(config(P, TRACKS5), params("composer", enc="DBN", enc_hidden=[10, 8], gen="NADE", Hn=16, units=[32, 32]), mode="composer", precision="fp32")
    a.train_step(x, None, AdamOptimizer(0.01))
    a.check()
    for store in [a.generators[0].store, a.encoders[0].store]:
        store.skipped.fill_(1)
        with pytest.raises(FloatingPointError):
            a.check()
        a.check()                                        # the counter is cleared by the raise
    f = MultINN(config(P, TRACKS5), params("feedback-rnn", gen="NADE", Hn=16, units=[32, 32], feedback=[64, 32]), mode="feedback-rnn", precision="fp32")
    f.train_step(x, None, AdamOptimizer(0.01))
    f.check()
    f._feedback_layer.store.skipped.fill_(1)
    with pytest.raises(FloatingPointError):
        f.check()


def test_load_encoders_finds_another_mode_class_checkpoint_and_restores_weights_only(tmp_path):
    """multinn_core.py:425-448 restores the encoder variables from whatever checkpoint the directory holds: encoders pre-trained under one
    mode name are found by a model constructed under another; only the weights are copied (the Adam slots of the pre-training stay behind)."""
    from multinn_amd import MultINN
    P = 12
    kw = dict(enc="DBN", enc_hidden=[10, 8], gen="NADE", Hn=16, units=[32, 32])
    a = MultINN(config(P, TRACKS5), params("composer", **kw), mode="composer", name="pretrain-run", precision="fp32")
    for e in a.encoders:
        e.store.theta.add_(torch.randn_like(e.store.theta) * 0.1)
        e.store.m.fill_(3.0)
        e.store.step_dev.fill_(5)
    a.save(None, str(tmp_path))
    b = MultINN(config(P, TRACKS5), params("composer", **kw), mode="composer", precision="fp32")
    assert b.name != a.name and b.load_encoders(None, str(tmp_path)) is True
    for ea, eb in zip(a.encoders, b.encoders):
        assert torch.equal(ea.store.theta, eb.store.theta) and float(eb.store.m.abs().max()) == 0.0 and int(eb.store.step_dev) == 0


@pytest.mark.parametrize("mode,gen,enc,enc_hidden", [("jamming", "RBM", "Pass", None), ("composer", "NADE", "DBN", [12, 8]), ("jamming", "NADE", "Pass", None)])
def test_mode_ragged_step_is_captured_once_for_any_lengths(mode, gen, enc, enc_hidden):
    """MultINNCore.graphed_train_step(lengths=...): ONE captured hipGraph of a mode's train step -- encoders, every generator's build and
    backward, the joint clipped step -- serves every later (x, lengths): row weights, valid-row counts, the f16 loss scale and (NADE generators)
    the compaction of the rows are derived on the device inside the graph (`RnnEstimator.ragged_on_device`).  Three replays on different
    length vectors (one full-length, one nearly empty) against eager steps of a twin model on the same data: the same losses and weights."""
    from multinn_amd import MultINN, AdamOptimizer
    B, T, P = 32, 8, 8
    tracks = TRACKS5[:3]
    rng = np.random.default_rng(4)
    xs = [dev(batch(B, T, P, len(tracks), 30 + i, rho=0.15)) for i in range(3)]
    lens = [rng.integers(1, T + 2, B).astype(np.int32), np.full(B, T + 1, np.int32), np.ones(B, np.int32)]
    lens[2][0] = 4

    def model():
        m = MultINN(config(P, tracks), params(mode, enc=enc, enc_hidden=enc_hidden, gen=gen, Hn=256, units=(128, 128)), mode=mode, precision="fp16", seed=23)
        return m, AdamOptimizer(0.01)

    (a, oa), (b, ob) = model(), model()
    b.train_step(xs[0], dev(lens[0]), ob)                       # materialises the twin (its weights are copied from a below)
    a.train_step(xs[0], dev(lens[0]), oa)
    for ga, gb in zip(a.generators, b.generators):
        gb.store.theta.copy_(ga.store.theta); gb.store.m.copy_(ga.store.m); gb.store.v.copy_(ga.store.v)
        gb.store.step = ga.store.step
        gb.store.step_dev.copy_(ga.store.step_dev)
        gb._packed_step = -1
    for ea, eb in zip(a.encoders, b.encoders):
        if getattr(ea, "store", None) is not None and ea.store.theta is not None:
            eb.store.theta.copy_(ea.store.theta)
    run = a.graphed_train_step(xs[0], oa, warmup=1, lengths=dev(lens[0]))
    assert run.ragged
    for ga, gb in zip(a.generators, b.generators):             # the capture's warm-up step ran on a: replay it on the twin
        pass
    b.train_step(xs[0], dev(lens[0]), ob)
    for x, ln in zip(xs, lens):
        la = float(run(x, dev(ln)))
        lb = float(b.train_step(x, dev(ln), ob))
        assert abs(la - lb) < 2e-3 * max(1.0, abs(lb)), (la, lb)
    a.check()
    b.check()
    for ga, gb in zip(a.generators, b.generators):
        assert float((ga.store.theta - gb.store.theta).abs().max()) < 5e-3


def test_driver_epoch_replays_captured_mode_steps(monkeypatch):
    """driver.train_epoch on a MODE class (jamming: three LSTM-RBM generators): windows of a recurring shape -- full-length and ragged -- run as
    replays of the mode's captured step (captured at the shape's second occurrence with warmup = 0, so nothing extra executes): the loss
    trajectory of two epochs is the eager loop's (MULTINN_TRAIN_GRAPH=0)."""
    from multinn_amd import MultINN, AdamOptimizer
    from multinn_amd.driver import train_epoch, LossAccumulator, TrainingStats
    R = np.random.default_rng(6)
    P, tracks = 8, TRACKS5[:3]
    X = (R.random((16, 12, P, len(tracks))) < .2).astype(np.uint8)
    lengths = np.full(16, 12)
    lengths[3] = 7
    ids = np.arange(16)

    def epoch_losses(graph):
        if not graph:
            monkeypatch.setenv("MULTINN_TRAIN_GRAPH", "0")
        m = MultINN(config(P, tracks), params("jamming", gen="RBM", Hn=32, units=(128, 128)), mode="jamming", precision="fp16", seed=23)
        opt = AdamOptimizer(0.01)
        out = []
        for _ in range(3):
            acc = LossAccumulator()
            train_epoch(m, X, lengths, ids, 8, 4, opt, acc, TrainingStats(), lr=0.01, device=DEV)
            out.append(acc.loss())
        if not graph:
            monkeypatch.delenv("MULTINN_TRAIN_GRAPH")
        m.check()
        return out, m

    lg, mg = epoch_losses(True)
    le, me = epoch_losses(False)
    keys = list(mg.__dict__.get("_step_graphs", {}))
    assert any(k[-1] == "ragged" for k in keys) and any(k[-1] == "full" for k in keys), keys
    assert "_step_graphs" not in me.__dict__
    assert np.allclose(lg, le, rtol=2e-2), (lg, le)
