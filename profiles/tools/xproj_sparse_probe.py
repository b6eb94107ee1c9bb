"""Layer 1's input projection at the bench shape ([262144 x 448] binary input -> 2048 gate columns, f16): the sum of the ON notes' weight rows
(mnn_xproj_sparse) against the dense product (mnn_gemm_tn), one process, interleaved.   python profiles/tools/xproj_sparse_probe.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from multinn_amd import ops

N, K, ncols = 262144, 448, 2048
Wd = (torch.randn(ncols, K, device="cuda") * 0.1).half()
Wt = Wd.t().contiguous()
bias = torch.randn(ncols, device="cuda")
for rho in (0.03, 0.07, 0.15):
    x = (torch.rand(N, K, device="cuda") < rho)
    x[:, 440:] = False
    mask = torch.from_numpy(np.packbits(x.cpu().numpy().astype(np.uint8), axis=1, bitorder="little")).cuda()
    xh = x.half()
    o1 = torch.empty(N, ncols, device="cuda", dtype=torch.float16)
    o2 = torch.empty_like(o1)
    ts = {"sparse": [], "gemm": []}
    for rd in range(6):
        for kind in ("sparse", "gemm"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                if kind == "sparse":
                    ops.xproj_sparse(mask, Wt, bias, o1)
                else:
                    ops.gemm_tn(xh, Wd, o2, bias=bias)
            e1.record()
            torch.cuda.synchronize()
            if rd:
                ts[kind].append(e0.elapsed_time(e1) / 3)
    d = float((o1.float() - o2.float()).abs().max())
    print(f"rho={rho}: sum of ON rows {sorted(ts['sparse'])[2]*1e3:.0f} us | dense product {sorted(ts['gemm'])[2]*1e3:.0f} us | max |diff| {d:.2e}", flush=True)
