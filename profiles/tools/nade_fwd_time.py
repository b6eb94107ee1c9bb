"""Times the fp16 mode's NADE forward (split-operand matrix-core form) at the bench shape through the generator (N = 262 144 rows, D = 440, Hn = 256):
per-call HIP events of 5 eager train-step forwards.   python profiles/tools/nade_fwd_time.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from multinn_amd import _lib, RnnNade

B, T, P, M = 1024, 256, 88, 5
rng = np.random.default_rng(5)
x = torch.from_numpy((rng.random((B, T + 1, P, M)) < 0.03).astype(np.uint8)).cuda()
gen = RnnNade(P * M, 256, [512, 256], keep_prob=0.9, precision="fp16", seed=3)
for it in range(6):
    if it == 1:
        _lib.TIMING = {}
    gen.build_pianoroll(x, None, is_train=True, mode="train")
    gen.backward()
torch.cuda.synchronize()
t = _lib.TIMING
_lib.TIMING = None
for k, v in sorted(t.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    ms = sorted(a.elapsed_time(b) for a, b in v)
    if ms[len(ms) // 2] > 0.1:
        print(f"{k:34s} median {ms[len(ms) // 2]:.3f} ms  (min {ms[0]:.3f}, {len(ms)} calls)")
