"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle cannot run
[256,128,88,5] in seconds): closed forms, permutation / batch-split invariance (the data-parallel identity),
sample <-> log_prob consistency, determinism, fp32 vs bf16 agreement.  All through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
B, T, P, M, HN, UNITS = 256, 128, 88, 5, 256, [512, 256]        # BASELINE.json configs[1]
D = P * M


def synth(B_, T_, seed, rho=0.03):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy((rng.random((B_, T_, P, M)) < rho).astype(np.uint8)).to(DEV)


def test_K1_zero_weight_closed_form_at_full_size():
    """NADE with zero weights and biases: p = 0.5, NLL = -D log(0.500001) for every one of the 32768 rows."""
    from multinn_amd import ops
    N = B * T
    v = synth(B, T, 1).view(1, N, D)
    z = lambda *s: torch.zeros(s, device=DEV)
    nll, cp = z(1, N), z(1, N, D)
    ops.nade_logprob_fwd(v, z(N, HN + D), z(1, D, HN), z(1, D, HN), 1, D, HN, None, nll, cp)
    assert float((cp - 0.5).abs().max()) < 1e-7
    assert float((nll - 304.98388).abs().max()) < 1e-3


def test_lstm_zero_weights_full_size():
    """K12 at full size: W = 0, b = 0 -> h = c = 0 for all 128 steps (bf16 step kernels, units 512)."""
    from multinn_amd import ops
    u = 512
    xproj = torch.zeros((T, B, 4 * u), device=DEV)
    wh = torch.zeros((4 * u, u), device=DEV, dtype=torch.bfloat16)
    c = torch.ones((T, B, u), device=DEV); h = torch.ones((T, B, u), device=DEV, dtype=torch.bfloat16)
    ops.lstm_seq_fwd(xproj, wh, None, None, None, c, h)
    assert float(c.abs().max()) == 0 and float(h.float().abs().max()) == 0


@pytest.fixture(scope="module")
def gen():
    from multinn_amd import RnnNade
    g = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="bf16", seed=23)
    g._materialize(D)
    return g


def test_full_size_step_is_finite_deterministic_and_permutation_invariant(gen):
    x = synth(B, T, 2)
    gen.row0 = 0
    gen.build_pianoroll(x, None, True, "train")
    l0 = float(gen.metrics["batch/loss"])
    nll0 = gen._nll_tm.clone()
    assert np.isfinite(l0) and 50 < l0 < 400                       # random init: about D * 0.69 * (small-p correction)
    gen.backward()
    g0 = gen.store.grad.clone()
    assert bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    gen.build_pianoroll(x, None, True, "train")
    assert torch.equal(gen._nll_tm, nll0)                          # bit-deterministic forward
    # without dropout the loss does not depend on the order of the sequences in the batch
    gen._rnn._keep_prob = 1.0
    gen.build_pianoroll(x, None, True, "train")
    la = float(gen.metrics["batch/loss"])
    perm = torch.randperm(B, device=DEV)
    gen.build_pianoroll(x[perm].contiguous(), None, True, "train")
    lb = float(gen.metrics["batch/loss"])
    gen._rnn._keep_prob = 0.9
    assert abs(la - lb) < 2e-5 * abs(la)
    # and with dropout the per-sequence results follow the GLOBAL sequence index (row0), not the position
    gen.row0 = 0
    gen.build_pianoroll(x, None, True, "train")
    full = gen._nll_tm.view(T, B).clone()
    gen.row0 = B // 2
    gen.build_pianoroll(x[B // 2:].contiguous(), None, True, "train")
    half = gen._nll_tm.view(T, B // 2)
    gen.row0 = 0
    assert torch.allclose(half, full[:, B // 2:], rtol=1e-4, atol=1e-3)


def test_batch_split_gradient_identity(gen):
    """The data-parallel identity on the GPU: grad(full batch) == (grad(half 1) + grad(half 2)) / 2 with the halves'
    RNG keyed by the global sequence index -- what one all-reduce over two ranks computes."""
    x = synth(64, 32, 3)
    gen.row0 = 0
    gen.build_pianoroll(x, None, True, "train"); gen.backward()
    g_full, l_full = gen.store.grad.clone(), float(gen.metrics["batch/loss"])
    gen.build_pianoroll(x[:32].contiguous(), None, True, "train"); gen.backward()
    g1, l1 = gen.store.grad.clone(), float(gen.metrics["batch/loss"])
    gen.row0 = 32
    gen.build_pianoroll(x[32:].contiguous(), None, True, "train"); gen.backward()
    g2, l2 = gen.store.grad.clone(), float(gen.metrics["batch/loss"])
    gen.row0 = 0
    assert abs(l_full - 0.5 * (l1 + l2)) < 1e-4 * abs(l_full)
    ref = 0.5 * (g1 + g2)
    assert float((g_full - ref).abs().max()) < 2e-2 * float(ref.abs().max())      # bf16 GEMM operands, f32 atomics


def test_fp32_and_bf16_paths_agree_at_scale():
    from multinn_amd import RnnNade
    x = synth(64, 64, 4)
    a = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="fp32", seed=7)
    b = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="bf16", seed=7)
    a._materialize(D); b._materialize(D)
    b.store.theta.copy_(a.store.theta)
    a.build_pianoroll(x, None, True, "train"); b.build_pianoroll(x, None, True, "train")
    la, lb = float(a.metrics["batch/loss"]), float(b.metrics["batch/loss"])
    assert abs(la - lb) < 5e-3 * abs(la)
    a.backward(); b.backward()
    cos = torch.nn.functional.cosine_similarity(a.store.grad, b.store.grad, dim=0)
    assert float(cos) > 0.999


def test_sampling_consistency_at_default_sampling_batch(gen):
    """A10/A11 at the reference's sampling batch (72 = 24 intros x 3, default_config.yaml:43-51): the NLL returned by
    the sampling kernel equals log_prob of its own sample; the scan is reproducible."""
    from multinn_amd import ops
    n = 72
    g = torch.Generator(device=DEV).manual_seed(5)
    bias = torch.randn((n, 704), device=DEV, generator=g) * 0.5
    we, wd = gen.store["nade/w_enc"], gen.store["nade/w_dec"]
    smp = torch.empty((n, D), device=DEV, dtype=torch.uint8); nll_s = torch.empty((1, n), device=DEV)
    ops.nade_sample(bias, we, wd, 1, D, HN, 1.0, 11, 0, 0, smp, nll=nll_s)
    nll = torch.empty((1, n), device=DEV); cp = torch.empty((1, n, D), device=DEV)
    ops.nade_logprob_fwd(smp.view(1, n, D), bias, we, wd, 1, D, HN, None, nll, cp)
    assert torch.allclose(nll, nll_s, rtol=2e-5, atol=1e-3)
    intro = synth(n, 8, 6).view(n, 8, D)
    s1 = gen.generate(intro, 4)
    assert s1.shape == (n, 4, D) and torch.equal(s1, gen.generate(intro, 4))
    assert 0 < float(s1.float().mean()) < 1


def test_ragged_and_empty_sequences(gen):
    """Edge cases of train.py:166-173: ragged lengths including a zero-length sequence contribute zero weight."""
    x = synth(8, 16, 8)
    lengths = torch.tensor([16, 0, 5, 16, 1, 9, 16, 3], dtype=torch.int32, device=DEV)
    gen.build_pianoroll(x, lengths, False, "eval")
    assert gen.log_probs.shape[0] == int(lengths.sum())
    l_all = float(gen.metrics["batch/loss"])
    keep = lengths > 0
    gen.build_pianoroll(x[keep].contiguous(), lengths[keep].contiguous(), False, "eval")
    assert abs(float(gen.metrics["batch/loss"]) - l_all) < 1e-5 * abs(l_all)


def test_persistent_recurrence_is_race_free_under_load():
    """Hand-off protocol of the persistent LSTM launches (flags + tiles between workgroups): 40 forward + backward passes at
    C2 size on the same parameters must reproduce the per-row NLL and the transposed dz of both layers BIT FOR BIT (a stale or
    early tile read shows up as run-to-run noise; everything downstream of f32 atomics is excluded), also while a second
    stream keeps the chip unevenly busy; no launch may have given up on a spin."""
    from multinn_amd import RnnNade
    x = synth(B, 64, 31)
    g = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="bf16", seed=5)
    g._materialize(D)
    g._stack.keep_debug = True
    noise = torch.empty((64, 1 << 20), device=DEV)
    side = torch.cuda.Stream()
    ref = None
    for i in range(40):
        if i % 3 != 0:
            with torch.cuda.stream(side):
                noise[: 8 + (i % 5) * 8].mul_(1.0001)              # a few CUs' worth of unrelated streaming work, off and on
        g.build_pianoroll(x, None, True, "train")
        nll = g._nll_tm.clone()
        g.backward()
        dz = [d.clone() for d in g._stack._dbg_dzT]
        if ref is None:
            ref = (nll, dz)
            assert bool(torch.isfinite(nll).all()) and all(bool(torch.isfinite(d.float()).all()) for d in dz)
        else:
            assert torch.equal(nll, ref[0]), i
            assert all(torch.equal(a_, b_) for a_, b_ in zip(dz, ref[1])), i
    torch.cuda.synchronize()
    assert g._stack._persist(B)
    g._stack.check()


def test_two_graph_step_replays_at_full_size(monkeypatch):
    """Data-parallel form of the captured step at C2 size: forward+backward graph | clip+Adam graph, five replays, no
    persistent launch may give up on a spin (see test_bench_data_parallel_rehearsal_over_rccl for the failure this guards)."""
    from multinn_amd import RnnNade, AdamOptimizer
    import multinn_amd.generators as gmod
    monkeypatch.setattr(gmod, "dp_active", lambda: True)
    x = synth(B, 128, 37)
    g = RnnNade(D, HN, UNITS, keep_prob=0.9, precision="bf16", seed=5)
    opt = AdamOptimizer(0.01)
    for _ in range(2):                                  # as bench.py does: eager steps on the main stream first
        g.train_step(x, None, opt)
    run = g.graphed_train_step(x, opt, warmup=1)
    losses = []
    for _ in range(5):
        losses.append(float(run()))
        g._stack.check()                                # also a blocking device-to-host copy between the replays, as the failing run had
    assert all(np.isfinite(losses)) and max(losses) < 1.5 * losses[0] and min(losses) > 0.5 * losses[0], losses
    assert g.store.step == 8 and int(g.store.step_dev) == 8


@pytest.mark.parametrize("comm", ["torch", "capi"])
def test_bench_data_parallel_rehearsal_over_rccl(comm):
    """(comm = capi: the all-reduce between the two graphs goes through the library's own mnn_allreduce_flat -- MULTINN_COMM=capi, a RCCL
    communicator created through the C ABI, the unique id broadcast as an object -- instead of torch.distributed.all_reduce.)
    bench.py's N>1 code path on this one GPU: torch.distributed.run with ONE rank, backend nccl (= RCCL), and
    MULTINN_DP_REHEARSAL=1 so that the step takes the two-graph form with the eager all-reduce of the flat gradient between
    them.  With hipMemsetAsync nodes in the graphs this run failed on the second replay (stale 16-byte fill pattern over the
    hand-off flags -> 'persistent LSTM launch timed out'); the library now fills with its own kernels."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:                          # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MULTINN_DP_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if comm == "capi":
        env["MULTINN_COMM"] = "capi"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["launch"] == "hipgraph-replay" and out["value"] > 0
    assert out["dp"].get("collective") == ("mnn_allreduce_flat (C ABI)" if comm == "capi" else "torch.distributed.all_reduce")
    smp = out["sampling"]                      # the sampling scan runs on rank 0 after the timed region, process group still up
    assert smp["unit"] == "generated timesteps/s" and smp["value"] > 0 and smp["launch"] == "hipgraph-replay"
