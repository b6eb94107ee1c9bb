"""The weight-resident GEMM (gemm_bres.hip) against the LDS-staged forms on the step's input projections, one process, interleaved:
MNN_GEMM_BRES=0 (256 x 256 / pair kernels) | 1 (weight-resident).  Checked against an f32 product of the same 16-bit operands first.
    python profiles/tools/gemm_bres_probe.py"""
import os
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops
dev = "cuda"


def timed(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K, dt, hb) in [(8192, 2048, 448, torch.float16, True), (16384, 1024, 512, torch.bfloat16, True), (8192 + 128 * 5, 256, 448, torch.float16, False), (8192, 264, 448, torch.float16, True),
                          (32768, 512, 512, torch.float16, True), (8192 * 3 + 128, 256, 512, torch.float16, False)]:
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(dt)
    Bm = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(dt)
    bias = torch.randn(N, device=dev, generator=g) if hb else None
    ref = A.float() @ Bm.float().t() + (bias if hb else 0.0)
    ldc = (N + 63) // 64 * 64
    for mode in ("0", "1"):
        os.environ["MNN_GEMM_BRES"] = mode
        Cfull = torch.full((M, ldc), 7.0, device=dev, dtype=dt)
        ops.gemm_tn(A, Bm, Cfull[:, :N], bias=bias)
        torch.cuda.synchronize()
        err = float((Cfull[:, :N].float() - ref).abs().max())
        ok_pad = bool((Cfull[:, N:] == 7.0).all())
        print(f"check M={M} N={N} K={K} {str(dt)[6:]} mode {mode}: max err {err:.3e} pad untouched {ok_pad}", flush=True)
        assert err < (0.25 if dt == torch.bfloat16 else 0.03) and ok_pad

Nr = 262144
bufs = []
for name, N, K in (("xproj1", 2048, 448), ("xproj2", 1024, 512)):
    A = (torch.randn(Nr, K, device=dev) * 0.5).half()
    Bm = (torch.randn(N, K, device=dev) * 0.1).half()
    C = torch.empty(Nr, N, device=dev, dtype=torch.float16)
    bufs.append((name, A, Bm, C, torch.randn(N, device=dev)))
res = {}
MODES = ("0", "1", "1v1", "1v2", "1v3")      # 1vN: development variant N of the weight-resident kernel (gemm_bres.hip)
for rd in range(6):
    for name, A, Bm, C, bias in bufs:
        for mode in MODES:
            os.environ["MNN_GEMM_BRES"] = mode[0]
            os.environ["MNN_GEMM_BRES_VAR"] = mode[2:] if len(mode) > 1 else "0"
            t = timed(lambda: ops.gemm_tn(A, Bm, C, bias=bias))
            if rd:
                res.setdefault((name, mode), []).append(t)
for name, A, Bm, C, bias in bufs:
    fl = 2.0 * A.shape[0] * A.shape[1] * Bm.shape[0]
    line = f"{name}: "
    for mode in MODES:
        ts = sorted(res[(name, mode)])
        line += f"mode {mode} {ts[len(ts)//2]*1e3:.1f} us (min {ts[0]*1e3:.1f}) {fl/ts[len(ts)//2]/1e9:.0f} TF/s | "
    print(line, flush=True)
