"""NADE backward: the matrix-core form (csrc/nade_bwd2.hip, `f16_products`) against the vector scan on the same loss-scaled inputs, in one process:
results compared, both timed at the bench shapes.      python profiles/tools/nade_bwd2_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops


def setup(N, D, Hn, tracks, rho, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = (torch.rand((tracks, N, D), device="cuda", generator=g) < rho).to(torch.uint8)
    ld = (tracks * (Hn + D) + 63) // 64 * 64
    bias = (torch.randn((N, ld), device="cuda", generator=g) * 0.5)[:, :tracks * (Hn + D)]
    we = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    wd = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    rw = torch.rand(N, device="cuda", generator=g) * 256.0
    d0 = torch.zeros((N, ld), device="cuda")[:, :tracks * (Hn + D)]
    af = torch.zeros((tracks, N, Hn), device="cuda")
    ops.nade_logprob_fwd(v, bias, we, wd, tracks, D, Hn, rw, torch.zeros((tracks, N), device="cuda"), None, d0, af)
    return v, bias, we, wd, d0, af


def run(f16, v, bias, we, wd, d0, af, tracks, D, Hn):
    dwe, dwd = torch.zeros_like(we), torch.zeros_like(wd)
    d1 = torch.zeros_like(d0.as_strided((d0.shape[0], d0.stride(0)), (d0.stride(0), 1)))[:, :d0.shape[1]]
    d1.copy_(d0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.nade_logprob_bwd(v, bias, we, wd, tracks, D, Hn, af, d1, dwe, dwd, f16_products=f16)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), dwe, dwd, d1[:, :tracks * Hn].clone()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


for (N, D, Hn, tracks) in [(262144, 440, 256, 1), (262144, 88, 128, 5), (32768, 440, 256, 1)]:
    for rho in (0.03, 0.1, 0.3):
        ins = setup(N, D, Hn, tracks, rho)
        tv, tm = [], []
        for rd in range(5):
            t, e0, d0_, b0 = run(False, *ins, tracks, D, Hn)
            tv.append(t)
            t, e1, d1_, b1 = run(True, *ins, tracks, D, Hn)
            tm.append(t)
        print(f"N={N} D={D} Hn={Hn} tracks={tracks} rho={rho}: vector scan {sorted(tv)[2]:.3f} ms | matrix cores {sorted(tm)[2]:.3f} ms   "
              f"rel diff d w_enc {rel(e1, e0):.2e}  d w_dec {rel(d1_, d0_):.2e}  d b_enc {rel(b1, b0):.2e}", flush=True)
