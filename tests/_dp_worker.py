"""Worker of tests/test_gpu_dp.py: one data-parallel rank of the PRODUCT train step (run under torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    precision, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    kind = sys.argv[4] if len(sys.argv) > 4 else "nade"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                      # every rank on the one GPU of the box: gloo moves the gradient (RCCL refuses duplicate devices)
    dist.init_process_group("gloo")
    from multinn_amd import RnnNade, AdamOptimizer
    from multinn_amd.common import RBM, ParamStore
    B, T, P, M = 8, 6, 8, 2
    D = P * M
    x = (np.random.default_rng(3).random((B, T, P, M)) < 0.25).astype(np.uint8)
    lengths = np.array([6, 3, 5, 6, 2, 6, 4, 1], np.int32)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    dev = "cuda:0"
    if kind == "nade":
        gen = RnnNade(D, 16, [32, 32], keep_prob=0.9, precision=precision, seed=5)
        gen.row0 = rank * per
        opt = AdamOptimizer(0.01)
        losses = []
        for s in range(steps):
            ragged = s % 2 == 1                       # full-length and ragged windows: row weights 1/N_global either way
            l = gen.train_step(torch.from_numpy(x[sl]).to(dev), torch.from_numpy(lengths[sl]).to(dev) if ragged else None, opt)
            lt = l.clone()
            n_loc = float(lengths[sl].sum()) if ragged else float(per * T)
            dist.all_reduce(lt)                       # the logged loss: sum of the ranks' weighted partial sums
            losses.append(float(lt))
        if rank == 0:
            torch.save({"theta": gen.store.theta.cpu(), "losses": losses, "step": gen.store.step}, out_path)
    else:                                             # CD-k update of an RBM: the flat [dW | dbv | dbh] delta is all-reduced
        store = ParamStore(torch.device(dev))
        rbm = RBM(D, 12, k=2)
        rbm.declare(store, torch.Generator().manual_seed(1))
        store.materialize()
        rbm.seed = 9
        v = torch.from_numpy(x.reshape(B * T, D)).to(dev)
        rows = B * T // world
        rbm.visible_bias_init_ops(v[rank * rows:(rank + 1) * rows])[0]()
        bv0 = rbm.bv.clone().cpu()
        rbm.train(v[rank * rows:(rank + 1) * rows].contiguous(), 0.1, row0=rank * rows, sub0=0)
        if rank == 0:
            torch.save({"theta": store.theta.cpu(), "bv0": bv0}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
