"""Stage clocks of the cluster recurrence (lstm_cluster.hip built with -DCL_TRACE: python profiles/tools/build_trace_lib.py CL_TRACE scratch/lib_cl_trace.so):
wall-clock stamps (100 MHz) of wave 0 of workgroup 0 per timestep.
    MULTINN_HIP_LIB=scratch/lib_cl_trace.so python profiles/tools/cluster_trace.py [B] [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ.setdefault("MULTINN_HIP_LIB", "scratch/lib_cl_trace.so")
from multinn_amd import ops, _lib   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
u, dev, dt = 512, "cuda", torch.float16
g = torch.Generator(device="cpu").manual_seed(5)
wh = (torch.randn((4 * u, u), generator=g) * 0.04).to(dev).to(dt)
xproj = (torch.randn((T, B, 4 * u), generator=g) * 1.5).to(dev).to(dt)
mask = (torch.rand((T, B, u), generator=g) < 0.9).to(torch.uint8).to(dev)
N = T * B
bufs = dict(gates=torch.zeros((T, B, 4 * u), device=dev, dtype=dt), c=torch.zeros((T, B, u), device=dev), h=torch.zeros((T, B, u), device=dev, dtype=dt),
            hT=torch.zeros((u, N), device=dev, dtype=dt), yT=torch.zeros((u, N), device=dev, dtype=dt), y=torch.zeros((T, B, u), device=dev, dtype=dt))
ws = ops.lstm_rowpar_workspace(T, B, u, dev)
d = ops.lstm2_fwd_layer(xproj, wh, None, None, bufs["gates"], bufs["c"], bufs["h"], bufs["hT"], bufs["y"], mask, yT=bufs["yT"], gates_dtype=dt, xproj_dtype=dt)
for _ in range(3):
    ops.lstm_cluster_fwd(T, B, d, 0.9, ws)
torch.cuda.synchronize()
ops.lstm_rowpar_check(ws)
lib = _lib.load()
buf = np.zeros((2, 512, 12), np.int64)
lib.mnn_lstm_cluster_trace.argtypes = [C.c_void_p]
assert lib.mnn_lstm_cluster_trace(buf.ctypes.data_as(C.c_void_p)) == 0


def show(title, st, names):
    seg = np.diff(st, axis=1) * 0.01
    step = np.diff(st[:, 0]) * 0.01
    print(f"{title}: median step {np.median(step):.3f} us (mean {np.mean(step):.3f})")
    for k, nm in enumerate(names):
        print(f"    {nm:44s} median {np.median(seg[:, k]):6.3f}  mean {np.mean(seg[:, k]):6.3f} us")


show("forward", buf[0, 8:min(T, 512) - 8, :6].astype(np.float64),
     ["wait for the cluster's 32 flags", "pull issue + staging issue + vmcnt + barrier", "MFMA (64) + pointwise (8 pairs)", "hand-off store -> vmcnt(0) -> flag", "output stores (issue)"])
dh = (torch.randn((T, B, u), generator=g) * 0.01).to(dev)
dzc = torch.zeros((T, B, 4 * u), device=dev, dtype=dt)
dzT = torch.zeros((N // 32, 4 * u, 32), device=dev, dtype=dt)
db = torch.zeros(4 * u, device=dev)
e = ops.lstm2_bwd_layer(dh, wh.t().contiguous(), bufs["gates"], bufs["c"], None, dzc, ops.lstm_seq_bwd_workspace(B, u, dev), dzT, db, mask, gates_dtype=dt)
for _ in range(3):
    ops.lstm_cluster_bwd(T, B, e, 0.9, ws)
torch.cuda.synchronize()
ops.lstm_rowpar_check(ws)
assert lib.mnn_lstm_cluster_trace(buf.ctypes.data_as(C.c_void_p)) == 0
if os.environ.get("MNN_CLUSTER_PIPE") == "0":
    show("backward", buf[1, 8:min(T, 512) - 8, :7].astype(np.float64),
         ["MFMA (64) + partial stores + output emit", "vmcnt(0) + flag", "wait for the cluster's 32 flags", "partial loads + sums", "pointwise + LDS writes", "operand requests + barrier"])
else:
    show("backward (two half tiles)", buf[1, 8:min(T, 512) - 8, :7].astype(np.float64),
         ["half 0: products + partial stores + vmcnt(0) + flag", "half 1 (previous step): poll + partial sums + pointwise", "barrier",
          "half 1: products + partial stores + vmcnt(0) + flag", "half 0: poll + partial sums + pointwise", "barrier"])
    rl = buf[1, 8:min(T, 512) - 8, 8:10].astype(np.float64)
    print(f"    granule reloads per step (wave 0 of workgroup 0): half 0 mean {rl[:, 0].mean():.2f} (max {rl[:, 0].max():.0f}), half 1 mean {rl[:, 1].mean():.2f} (max {rl[:, 1].max():.0f})")
