"""NADE backward (mnn_nade_logprob_bwd) of whatever library MULTINN_HIP_LIB points at, timed at the bench shape and checked against saved results:
    python profiles/tools/nade_bwd_lib_probe.py save|check <file.pt>
Run once with the product library (save), once with a variant library (check): same inputs (seeded), results compared, medians printed."""
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops

mode, path = sys.argv[1], sys.argv[2]


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


def run(v, bias, we, wd, d0, af, tracks, D, Hn):
    dwe, dwd = torch.zeros_like(we), torch.zeros_like(wd)
    d1 = torch.zeros_like(d0.as_strided((d0.shape[0], d0.stride(0)), (d0.stride(0), 1)))[:, :d0.shape[1]]
    d1.copy_(d0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.nade_logprob_bwd(v, bias, we, wd, tracks, D, Hn, af, d1, dwe, dwd)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), dwe, dwd, d1[:, :tracks * Hn].clone()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


saved = torch.load(path) if mode == "check" else {}
out = {}
for (N, D, Hn, tracks, rho) in [(300, 440, 256, 1, 0.03), (1000, 88, 100, 5, 0.1), (65, 7, 65, 3, 0.5), (262144, 440, 256, 1, 0.03), (262144, 440, 256, 1, 0.1),
                                (262144, 440, 256, 1, 0.5), (262144, 88, 128, 5, 0.03), (32768, 440, 256, 1, 0.03)]:
    ins = setup(N, D, Hn, tracks, rho)
    ts = []
    for _ in range(5 if N > 10000 else 1):
        t, dwe, dwd, db = run(*ins, tracks, D, Hn)
        ts.append(t)
    key = f"{N}_{D}_{Hn}_{tracks}_{rho}"
    msg = f"N={N} D={D} Hn={Hn} tracks={tracks} rho={rho}: {sorted(ts)[len(ts) // 2]:.3f} ms"
    if mode == "check":
        r = saved[key]
        msg += f"   vs saved: d w_enc {rel(dwe, r[0].cuda()):.2e}  d w_dec {rel(dwd, r[1].cuda()):.2e}  d b_enc {rel(db[:r[2].shape[0]], r[2].cuda()):.2e}"
    else:
        out[key] = (dwe.cpu(), dwd.cpu(), db.cpu()) if N <= 32768 else (dwe.cpu(), dwd.cpu(), db[:4096].cpu())
    if mode == "check" and N > 32768:
        pass
    print(msg, flush=True)
if mode == "save":
    torch.save(out, path)
