"""A few eager RAGGED train steps at the bench shape (lengths ~ U{T/2..T}, seed 24) for `rocprofv3 --kernel-trace --stats`:
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o ragged -- python3 profiles/tools/ragged_steps.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from multinn_amd import RnnNade, AdamOptimizer

B, T, P, M = 1024, 256, 88, 5
gen = RnnNade(P * M, bench.HN, bench.UNITS, keep_prob=0.9, precision="fp16", seed=23)
x = torch.from_numpy(bench.synth(B, T, P, M, 23, 0.03)).to("cuda")
ln = torch.from_numpy(bench.ragged_lengths(B, T)).to("cuda")
opt = AdamOptimizer(0.01)
for _ in range(4):
    gen.train_step(x, ln, opt)
torch.cuda.synchronize()
gen.check()
print("ok", int(ln.sum()), "valid rows of", B * T)
