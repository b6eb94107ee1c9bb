"""Stage clocks of the row-parallel persistent recurrence (lstm_rowpar.hip built with -DRP_TRACE into scratch/lib_rp_trace.so):
wall-clock stamps (100 MHz) of wave (member 0, row tile 0) per timestep: 0 item start | 1 hand-off flags seen | 2 loads + MFMA chain done |
3 gate pointwise done (tiles in LDS) | 4 hand-off stores issued | 5 row-major stores issued | 6 hand-off stores done, flag raised | 7 next operands requested.
Run from the repository root on the GPU box:  MULTINN_HIP_LIB=scratch/lib_rp_trace.so python profiles/tools/rowpar_trace.py [B] [T]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from multinn_amd import RnnNade, AdamOptimizer, _lib   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
UNITS = [int(u) for u in sys.argv[3].split(",")] if len(sys.argv) > 3 else [512, 256]
x = torch.from_numpy((np.random.default_rng(0).random((B, T, 88, 5)) < 0.03).astype(np.uint8)).cuda()
gen = RnnNade(440, 256, UNITS, keep_prob=0.9, precision=(sys.argv[4] if len(sys.argv) > 4 else "fp16"), seed=23)
opt = AdamOptimizer(0.01)
for _ in range(3):
    gen.train_step(x, None, opt)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((2, 512, 12), np.int64)
lib.mnn_lstm_rowpar_trace.argtypes = [C.c_void_p]
assert lib.mnn_lstm_rowpar_trace(buf.ctypes.data_as(C.c_void_p)) == 0
names = ["wait", "load+mfma", "pointwise", "xchg issue", "row-major stores", "drain+flag", "prefetch issue", "loop"]
for d, nm in enumerate((f"forward, layer {len(UNITS)} (units {UNITS[-1]})", f"backward, layer 1 (units {UNITS[0]})")):
    st = buf[d, 8:min(T, 512) - 8, :8].astype(np.float64)
    seg = np.diff(st, axis=1) * 0.01                       # us
    step = np.diff(st[:, 0]) * 0.01
    print(f"{nm}: median step {np.median(step):.2f} us")
    for k in range(7):
        print(f"    {names[k]:18s} {np.median(seg[:, k]):6.2f} us")
