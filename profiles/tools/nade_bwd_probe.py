"""(Needs the commit in which nade_bwd_mfma.hip was wired into the library: see the header of profiles/tools/nade_bwd_mfma.hip.)
NADE backward: the matrix-core scan (mnn_nade_logprob_bwd_mfma) against the vector scan (mnn_nade_logprob_bwd) on the same inputs, in one
process: small / ragged shapes first (results compared), then the bench shape (joint LSTM-NADE: N = 262 144 rows, D = 440, Hn = 256) timed.
    python profiles/tools/nade_bwd_probe.py [scale]        row weights = U[0,1) * scale: |d nll / d logit| <= scale (the fp16 mode's loss scale keeps it <= 256)"""
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 256.0


def setup(N, D, Hn, tracks, rho, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = (torch.rand((tracks, N, D), device="cuda", generator=g) < rho).to(torch.uint8)
    ld = (tracks * (Hn + D) + 63) // 64 * 64
    bias = (torch.randn((N, ld), device="cuda", generator=g) * 0.5)[:, :tracks * (Hn + D)]
    we = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    wd = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
    rw = torch.rand(N, device="cuda", generator=g) * scale
    d0 = torch.zeros((N, ld), device="cuda")[:, :tracks * (Hn + D)]
    af = torch.zeros((tracks, N, Hn), device="cuda")
    ops.nade_logprob_fwd(v, bias, we, wd, tracks, D, Hn, rw, torch.zeros((tracks, N), device="cuda"), None, d0, af)
    return v, bias, we, wd, d0, af


def run(kind, v, bias, we, wd, d0, af, tracks, D, Hn, wdp):
    dwe, dwd = torch.zeros_like(we), torch.zeros_like(wd)
    d1 = torch.zeros_like(d0.as_strided((d0.shape[0], d0.stride(0)), (d0.stride(0), 1)))[:, :d0.shape[1]]
    d1.copy_(d0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if kind == "vec":
        ops.nade_logprob_bwd(v, bias, we, wd, tracks, D, Hn, af, d1, dwe, dwd)
    else:
        ops.nade_logprob_bwd_mfma(v, we, wdp, tracks, D, Hn, af, d1, dwe, dwd)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), dwe, dwd, d1[:, :tracks * Hn].clone()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


for (N, D, Hn, tracks, rho) in [(256, 32, 32, 1, 0.05), (512, 88, 64, 2, 0.03), (300, 440, 256, 1, 0.03), (4096, 440, 256, 1, 0.03), (1024, 88, 128, 5, 0.1),
                                (2048, 440, 256, 1, 0.3)]:
    ins = setup(N, D, Hn, tracks, rho)
    wdp = ops.nade_bwd_pack(ins[3], tracks, D, Hn)
    _, e0, d0_, b0 = run("vec", *ins, tracks, D, Hn, wdp)
    _, e1, d1_, b1 = run("mfma", *ins, tracks, D, Hn, wdp)
    print(f"check N={N} D={D} Hn={Hn} tracks={tracks} rho={rho}: rel diff d w_enc {rel(e1, e0):.2e}  d w_dec {rel(d1_, d0_):.2e}  d b_enc {rel(b1, b0):.2e}", flush=True)

N, D, Hn, tracks = 262144, 440, 256, 1
for rho in (0.03, 0.1, 0.5):
    ins = setup(N, D, Hn, tracks, rho)
    wdp = ops.nade_bwd_pack(ins[3], tracks, D, Hn)
    tv, tm = [], []
    for rd in range(4):
        t, e0, d0_, b0 = run("vec", *ins, tracks, D, Hn, wdp)
        tv.append(t)
        t, e1, d1_, b1 = run("mfma", *ins, tracks, D, Hn, wdp)
        tm.append(t)
    print(f"rho={rho}: vector scan {sorted(tv[1:])[1]:.3f} ms | matrix-core scan {sorted(tm[1:])[1]:.3f} ms | rel diff d w_enc {rel(e1, e0):.2e} d w_dec {rel(d1_, d0_):.2e} "
          f"d b_enc {rel(b1, b0):.2e}", flush=True)
