"""NADE backward at the bench shape (joint LSTM-NADE: N = 262 144 rows, D = 440, Hn = 256), forms A/B in one process:
MNN_NADE_BWD_ACC=0 (cross-wave exchange per 4 visibles) / 1 (LDS accumulators per 8 visibles).  Outputs compared with each other
(sums in different orders: relative difference printed).  python profiles/tools/nade_bwd_probe.py"""
import os
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops

N, D, Hn, tracks = 262144, 440, 256, 1
for rho in (0.03, 0.5):
    g = torch.Generator(device="cuda").manual_seed(1)
    v = (torch.rand((tracks, N, D), device="cuda", generator=g) < rho).to(torch.uint8)
    bias = torch.randn((N, tracks * (Hn + D)), device="cuda", generator=g) * 0.5
    we = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    wd = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    rw = torch.rand(N, device="cuda", generator=g) / N
    z = lambda *s: torch.zeros(s, device="cuda")
    d0, af = torch.zeros_like(bias), z(tracks, N, Hn)
    ops.nade_logprob_fwd(v, bias, we, wd, tracks, D, Hn, rw, z(tracks, N), None, d0, af)
    outs = {}
    times = {"0": [], "1": []}
    for rd in range(4):
        for mode in ("0", "1"):
            os.environ["MNN_NADE_BWD_ACC"] = mode
            dwe, dwd, d1 = z(tracks, D, Hn), z(tracks, D, Hn), d0.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.nade_logprob_bwd(v, bias, we, wd, tracks, D, Hn, af, d1, dwe, dwd)
            e1.record()
            torch.cuda.synchronize()
            if rd > 0:
                times[mode].append(e0.elapsed_time(e1))
            outs[mode] = (dwe, dwd, d1[:, :tracks * Hn].clone())
    rel = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(outs["1"], outs["0"])]
    print(f"rho={rho}: exchange {sorted(times['0'])[1]:.3f} ms | LDS accumulators {sorted(times['1'])[1]:.3f} ms | max rel diff dwe {rel[0]:.2e} dwd {rel[1]:.2e} d b_enc {rel[2]:.2e}", flush=True)
