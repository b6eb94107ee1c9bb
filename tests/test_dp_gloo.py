"""Data-parallel path on CPU: world_size 2, gloo.  The batch dimension shards with ONE all-reduce
of the flat gradient.  Each rank runs the oracle on its shard with the SAME
conventions the product uses (row weight 1/N_global, dropout RNG keyed by the global sequence
index) and exchanges gradients through multinn_amd.training.allreduce_flat; the result must
equal the single-process full-batch gradient and optimiser step."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import generators as G


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(B=6, T=5, P=4, M=2, Hn=6, units=(8, 5)):
    R = np.random.Generator(np.random.PCG64(3))
    x = (R.random((B, T, P, M)) < .3).astype(np.float64)
    lengths = np.array([5, 3, 4, 2, 5, 1])
    p = G.init_rnn_nade(5, P * M, P * M, Hn, list(units), np.float64)
    return x, lengths, p, list(units)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multinn_amd.training import allreduce_flat, world as world_fn
    assert world_fn() == (rank, world)
    x, lengths, p, units = _setup()
    B = x.shape[0] // world
    sl = slice(rank * B, (rank + 1) * B)
    inp, tgt = G.joint_inputs(x[sl])
    n_local = torch.tensor([float(lengths[sl].sum())])
    n_tot = n_local.clone()
    dist.all_reduce(n_tot)                                     # what RnnEstimator._row_weight does
    du = G.dropout_uniforms(23, B, x.shape[1], units, row0=rank * B)   # RNG keyed by the GLOBAL sequence index
    fw = G.rnn_nade_forward(inp, tgt, lengths[sl], p, 0.9, du)
    g = G.rnn_nade_backward(fw, p, n_total=int(n_tot))
    flat = torch.from_numpy(np.concatenate([a.ravel() for a in G.flat_grads(g)]))
    allreduce_flat(flat)                                       # the ONE exchange of the step
    loss = torch.tensor([fw['loss'] * float(n_local) / float(n_tot)])
    dist.all_reduce(loss)
    if rank == 0:
        q.put((flat.numpy(), float(loss)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_equals_full_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    flat, loss = q.get(timeout=100)
    for pr in procs:
        pr.join(30)
        assert pr.exitcode == 0
    x, lengths, p, units = _setup()
    inp, tgt = G.joint_inputs(x)
    du = G.dropout_uniforms(23, x.shape[0], x.shape[1], units)
    fw = G.rnn_nade_forward(inp, tgt, lengths, p, 0.9, du)
    g = G.rnn_nade_backward(fw, p)
    ref = np.concatenate([a.ravel() for a in G.flat_grads(g)])
    assert np.allclose(flat, ref, rtol=1e-10, atol=1e-14)
    assert np.isclose(loss, fw['loss'], rtol=1e-12)


def test_single_process_world_is_identity():
    from multinn_amd.training import allreduce_flat, world
    assert world() == (0, 1)
    t = torch.arange(4.0)
    assert torch.equal(allreduce_flat(t.clone()), t)
