"""Stage clocks of the CU-resident recurrence (lstm_resident.hip built with -DRES_TRACE: profiles/tools/build_trace_lib.py RES_TRACE scratch/lib_res_trace.so):
wall-clock stamps (100 MHz) of wave 0 of workgroup 0 per timestep.
    MULTINN_HIP_LIB=scratch/lib_res_trace.so python profiles/tools/resident_trace.py [B] [T] [fwd|bwd]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ.setdefault("MULTINN_HIP_LIB", "scratch/lib_res_trace.so")
from multinn_amd import ops, _lib   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
u, dev, dt = 256, "cuda", torch.float16
g = torch.Generator(device="cpu").manual_seed(5)
wh = (torch.randn((4 * u, u), generator=g) * 0.06).to(dev).to(dt)
xproj = (torch.randn((T, B, 4 * u), generator=g) * 1.5).to(dev).to(dt)
mask = (torch.rand((T, B, u), generator=g) < 0.9).to(torch.uint8).to(dev)
N = T * B
bufs = dict(gates=torch.zeros((T, B, 4 * u), device=dev, dtype=dt), c=torch.zeros((T, B, u), device=dev), h=torch.zeros((T, B, u), device=dev, dtype=dt),
            hT=torch.zeros((u, N), device=dev, dtype=dt), yT=torch.zeros((u, N), device=dev, dtype=dt), y=torch.zeros((T, B, u), device=dev, dtype=dt))
d = ops.lstm2_fwd_layer(xproj, wh, None, None, bufs["gates"], bufs["c"], bufs["h"], bufs["hT"], bufs["y"], mask, yT=bufs["yT"], gates_dtype=dt, xproj_dtype=dt)
for _ in range(3):
    ops.lstm_resident_fwd(T, B, d, 0.9)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((2, 512, 12), np.int64)
lib.mnn_lstm_resident_trace.argtypes = [C.c_void_p]
assert lib.mnn_lstm_resident_trace(buf.ctypes.data_as(C.c_void_p)) == 0
names = ["emit + stage issue", "group 0", "group 1 + pw 0", "group 2 + pw 1", "group 3 + pw 2", "pw 3", "DMA wait", "barrier"]
st = buf[0, 8:min(T, 512) - 8, :9].astype(np.float64)
seg = np.diff(st, axis=1) * 0.01
step = np.diff(st[:, 0]) * 0.01
print(f"forward: median step {np.median(step):.3f} us")
for k in range(8):
    print(f"    {names[k]:22s} median {np.median(seg[:, k]):6.3f}  mean {np.mean(seg[:, k]):6.3f} us")
dh = (torch.randn((T, B, u), generator=g) * 0.01).to(dev)
dzc = torch.zeros((T, B, 4 * u), device=dev, dtype=dt)
dzT = torch.zeros((N // 32, 4 * u, 32), device=dev, dtype=dt)
db = torch.zeros(4 * u, device=dev)
e = ops.lstm2_bwd_layer(dh, wh.t().contiguous(), bufs["gates"], bufs["c"], None, dzc, ops.lstm_seq_bwd_workspace(B, u, dev), dzT, db, mask, gates_dtype=dt)
for _ in range(3):
    ops.lstm_resident_bwd(T, B, e, 0.9)
torch.cuda.synchronize()
assert lib.mnn_lstm_resident_trace(buf.ctypes.data_as(C.c_void_p)) == 0
st = buf[1, 8:min(T, 512) - 8, :4].astype(np.float64)
seg = np.diff(st, axis=1) * 0.01
print(f"backward: median step {np.median(np.diff(st[:, 0]) * 0.01):.3f} us")
for k, nm in enumerate(["MFMA phase (+ emit)", "pointwise", "requests + barrier"]):
    print(f"    {nm:22s} median {np.median(seg[:, k]):6.3f}  mean {np.mean(seg[:, k]):6.3f} us")
clk = np.diff(buf[0, 8:min(T, 512) - 8, 9].astype(np.float64))
print(f"shader clock: {np.median(clk / (step * 1e-6)) / 1e9:.3f} GHz (s_memtime ticks per wall-clock second over a step)")
