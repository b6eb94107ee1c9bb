"""Timing-only ablations of the two-half cluster backward (library built with -DCL_ABLATE: python profiles/tools/build_trace_lib.py CL_ABLATE scratch/lib_cl_abl.so;
MNN_CLUSTER_ABL bits: 1 no output emit, 2 no partial stores, 4 no granule loads, 8 no MFMAs, 16 no pointwise arithmetic; results are wrong with any of them).
    MULTINN_HIP_LIB=scratch/lib_cl_abl.so python profiles/tools/cluster_abl.py"""
import os
import sys
import torch
sys.path.insert(0, ".")
os.environ.setdefault("MULTINN_HIP_LIB", "scratch/lib_cl_abl.so")
from multinn_amd import ops

DEV = torch.device("cuda:0")
u, T, B, keep, dt = 512, 256, 1024, 0.9, torch.float16
g = torch.Generator(device="cuda").manual_seed(3)
wh_t = (torch.randn((4 * u, u), device=DEV, generator=g) * 0.04).to(dt)
mask = (torch.rand((T, B, u), device=DEV, generator=g) < keep).to(torch.uint8)
gates = torch.rand((T, B, 4 * u), device=DEV, generator=g).to(dt)
c = torch.randn((T, B, u), device=DEV, generator=g)
dh = torch.randn((T, B, u), device=DEV, generator=g) * 0.02
ws = ops.lstm_rowpar_workspace(T, B, u, DEV)
N = T * B
dzc = torch.zeros((T, B, 4 * u), device=DEV, dtype=dt)
dzT = torch.zeros((N // 32, 4 * u, 32), device=DEV, dtype=dt)
db = torch.zeros(4 * u, device=DEV)
E = ops.lstm2_bwd_layer(dh, wh_t.t().contiguous(), gates, c, None, dzc, ops.lstm_seq_bwd_workspace(B, u, DEV), dzT, db, mask, gates_dtype=dt)
for abl in (0, 1, 6, 7, 8, 15, 16, 31):
    os.environ["MNN_CLUSTER_ABL"] = str(abl)
    ts = []
    for rd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.lstm_cluster_bwd(T, B, E, keep, ws)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"ablate {abl:2d}: {sorted(ts[1:])[1]:.3f} ms ({sorted(ts[1:])[1] / T * 1e3:.2f} us/step)", flush=True)
