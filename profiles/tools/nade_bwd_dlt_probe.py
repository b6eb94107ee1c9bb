"""NADE backward with d nll / d logit read from a TRANSPOSED copy [track][visible][row] by scalar loads (library built with -DNB_DLT:
python profiles/tools/build_trace_lib.py NB_DLT scratch/lib_nb_dlt.so) against the shipped kernel (per-row scalars through v_readlane), one process each.
    MULTINN_HIP_LIB=scratch/lib_nb_dlt.so python profiles/tools/nade_bwd_dlt_probe.py dlt   |   python profiles/tools/nade_bwd_dlt_probe.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops, _lib

dlt = len(sys.argv) > 1 and sys.argv[1] == "dlt"


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


for (N, D, Hn, tracks) in [(4096, 440, 256, 1), (262144, 440, 256, 1), (262144, 88, 128, 5)]:
    for rho in (0.03, 0.1):
        ins = setup(N, D, Hn, tracks, rho)
        if dlt:
            d0 = ins[4]
            dlT = torch.stack([d0[:, tracks * Hn + m * D: tracks * Hn + (m + 1) * D].t().contiguous() for m in range(tracks)])     # [tracks, D, N]
            lib = _lib.load()
            lib.mnn_nade_bwd_set_dlt.argtypes = [C.c_void_p]
            assert lib.mnn_nade_bwd_set_dlt(C.c_void_p(dlT.data_ptr())) == 0
        ts = []
        for rd in range(5):
            t, dwe, dwd, b = run(*ins, tracks, D, Hn)
            ts.append(t)
        print(f"{'scalar loads of the transposed copy' if dlt else 'v_readlane (shipped)'}: N={N} D={D} Hn={Hn} tracks={tracks} rho={rho}: {sorted(ts[1:])[1]:.3f} ms   "
              f"checksums d w_enc {float(dwe.double().abs().sum()):.6e} d w_dec {float(dwd.double().abs().sum()):.6e} d b_enc {float(b.double().abs().sum()):.6e}", flush=True)
