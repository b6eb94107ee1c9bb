"""The PRODUCT data-parallel step with world size 2 on the one GPU of the box: two processes, each a rank with half of the batch,
gradient exchanged by ONE all-reduce of the flat buffer (gloo here -- RCCL refuses two ranks on one device; the call site is the
same `training.allreduce_flat`), against the single-process full-batch step.  Covers: 1/N_global row weights (full-length and
ragged), RNG keyed by global sequence index, clip on the REDUCED gradient, and the all-reduced CD delta of RBM.train."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def launch(world, args, extra_env=None):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MULTINN_TRAIN_GRAPH="0")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dp_worker.py")] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_rank_product_step_equals_full_batch_step(precision, tmp_path):
    from multinn_amd import RnnNade, AdamOptimizer
    out = str(tmp_path / "dp.pt")
    steps = 4
    launch(2, [precision, out, str(steps)])
    got = torch.load(out)
    B, T, P, M = 8, 6, 8, 2
    x = (np.random.default_rng(3).random((B, T, P, M)) < 0.25).astype(np.uint8)
    lengths = np.array([6, 3, 5, 6, 2, 6, 4, 1], np.int32)
    gen = RnnNade(P * M, 16, [32, 32], keep_prob=0.9, precision=precision, seed=5)
    opt = AdamOptimizer(0.01)
    losses = []
    for s in range(steps):
        losses.append(float(gen.train_step(torch.from_numpy(x).to(DEV), torch.from_numpy(lengths).to(DEV) if s % 2 else None, opt)))
    tol = 2e-5 if precision == "fp32" else 2e-2
    assert got["step"] == steps
    assert np.allclose(got["losses"], losses, rtol=tol), (got["losses"], losses)
    # Adam moves every weight by ~lr per step: a wrong weighting / clip / RNG split shows up as O(lr) differences
    assert float((got["theta"] - gen.store.theta.cpu()).abs().max()) < (2e-4 if precision == "fp32" else 4e-3)


def test_two_rank_cd_update_all_reduces_the_delta(tmp_path):
    from multinn_amd.common import RBM, ParamStore
    out = str(tmp_path / "cd.pt")
    launch(2, ["fp32", out, "1", "rbm"])
    got = torch.load(out)
    B, T, P, M = 8, 6, 8, 2
    D = P * M
    x = (np.random.default_rng(3).random((B, T, P, M)) < 0.25).astype(np.uint8)
    store = ParamStore(torch.device(DEV))
    rbm = RBM(D, 12, k=2)
    rbm.declare(store, torch.Generator().manual_seed(1))
    store.materialize()
    rbm.seed = 9
    v = torch.from_numpy(x.reshape(B * T, D)).to(DEV)
    rbm.visible_bias_init_ops(v)[0]()
    assert torch.allclose(got["bv0"], rbm.bv.cpu(), rtol=1e-5, atol=1e-6)          # p is the GLOBAL mean activation
    rbm.train(v, 0.1, row0=0, sub0=0)
    assert float((got["theta"] - store.theta.cpu()).abs().max()) < 1e-5              # same draws (global rows), N = global row count


def test_bench_two_ranks_on_one_gpu_over_gloo():
    """bench.py --gpus 2 started WITHOUT a rendezvous in the environment: the parent launches the ranks itself; both ranks share the
    device, the gradient travels over gloo.  The N > 1 product step (forward+backward graph | eager all-reduce | clip+Adam graph)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--steps", "3", "--warmup", "1",
           "--no-sampling", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0 and out["launch"] == "hipgraph-replay"
    assert out["roofline"]["step"]["frac"] > 0


def test_cabi_communicator_all_reduce_single_rank():
    """mnn_comm_unique_id / mnn_comm_init / mnn_allreduce_flat / mnn_comm_destroy: a one-rank RCCL communicator created
    through the C ABI alone (no torch.distributed), the flat f32 buffer all-reduced in place on the current stream -- with one rank the sum
    is the buffer itself -- and argument validation.  (More than one rank needs more than one GPU: the driver's scaling run.)"""
    import ctypes as C
    from multinn_amd import _lib
    from multinn_amd.training import CabiComm
    comm = CabiComm(0, 1)
    g = torch.arange(3_143_352, device="cuda:0", dtype=torch.float32) * 1e-3          # the joint model's flat gradient size
    ref = g.clone()
    comm.all_reduce(g)
    comm.all_reduce(g)
    torch.cuda.synchronize()
    assert torch.equal(g, ref)
    with pytest.raises(_lib.MnnError):
        _lib.call("mnn_allreduce_flat", None, None, C.c_void_p(g.data_ptr()), g.numel())
    with pytest.raises(_lib.MnnError):
        _lib.call("mnn_comm_init", C.byref(C.c_void_p()), 3, 2, C.create_string_buffer(128))
    comm.close()
    comm.close()
