"""[Historical: the pair kernel left libmultinn_hip.so in round 6 (gemm_pair_kernel.hip.frag); this probe needs the round-5 library.]
A/B of the short-K activation GEMMs of the bench-shape train step (TGT [1024,256,88,5]: 262 144 rows) in ONE process:
MNN_GEMM_PAIR=0 (the 256 x 256 kernel), =1 (pair kernel, C in whole lines through LDS), =2 (pair kernel, C straight from the accumulators).
Every form is first checked against an f32 torch product of the same 16-bit operands.  Run from the repository root on the GPU box:
    python profiles/tools/gemm_pair_probe.py [rounds]"""
import os
import sys
import torch
sys.path.insert(0, '.')
from multinn_amd import ops

dev = 'cuda'
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def timed(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


Nr = 262144
# name, M, N, K, C dtype, bias
shapes = [("xproj1", Nr, 2048, 448, torch.float16, True), ("xproj2", Nr, 1024, 512, torch.float16, True),
          ("dense fwd", Nr, 696, 256, torch.float32, True), ("dense dgrad", Nr, 256, 704, torch.float32, False),
          ("dX2", Nr, 512, 1024, torch.float32, False), ("dX2 f16 C", Nr, 512, 1024, torch.float16, False),
          ("dgrad f16 C", Nr, 256, 704, torch.float16, False)]
# correctness on small / edge shapes first (rows not a multiple of 256, N edge inside a wave tile, K = 64)
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K, cdt, hb) in [(1000, 696, 256, torch.float32, True), (512, 200, 64, torch.float16, True), (768, 2048, 448, torch.float16, True),
                           (300, 128, 96, torch.bfloat16, False), (4096, 512, 1024, torch.float32, False), (2048, 256, 704, torch.float32, True)]:
    for dt in (torch.float16, torch.bfloat16):
        if cdt != torch.float32 and cdt != dt:
            continue
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(dt)
        Bm = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(dt)
        bias = torch.randn(N, device=dev, generator=g) if hb else None
        ref = A.float() @ Bm.float().t() + (bias if hb else 0.0)
        ldc = (N + 63) // 64 * 64
        for mode in ("0", "1", "2"):
            os.environ["MNN_GEMM_PAIR"] = mode[0]
            os.environ["MNN_GEMM_PAIR_VAR"] = mode[2:] if len(mode) > 1 else "0"
            Cfull = torch.full((M, ldc), 7.0, device=dev, dtype=cdt)
            C = Cfull[:, :N]
            ops.gemm_tn(A, Bm, C, bias=bias)
            torch.cuda.synchronize()
            err = (C.float() - ref).abs().max().item()
            tol = 2e-3 if cdt == torch.float32 else (0.25 if cdt == torch.bfloat16 else 0.03)
            pad_ok = bool((Cfull[:, N:] == 7.0).all().item())
            print(f"check M={M} N={N} K={K} {str(dt)[6:]}->{str(cdt)[6:]} mode {mode}: max err {err:.3e} pad untouched {pad_ok}", flush=True)
            assert err < tol and pad_ok, "MISMATCH"

res = {}
# 0 = 256 x 256 kernel, 1 = pair kernel (whole-line epilogue), 2 = pair kernel (register epilogue); MvN = development variant N of mode M (gemm.hip)
MODES = ("0", "1", "2", "1v3", "1v4", "1v6", "1v7", "2v7", "1v8")
bufs = []
for name, M, N, K, cdt, hb in shapes:
    A = (torch.randn(M, K, device=dev) * 0.5).to(torch.float16)
    Bm = (torch.randn(N, K, device=dev) * 0.5).to(torch.float16)
    ldc = (N + 63) // 64 * 64
    C = torch.empty(M, ldc, device=dev, dtype=cdt)[:, :N]
    bias = torch.randn(N, device=dev) if hb else None
    bufs.append((name, A, Bm, C, bias))
for rd in range(rounds + 1):
    for name, A, Bm, C, bias in bufs:
        for mode in MODES:
            os.environ["MNN_GEMM_PAIR"] = mode[0]
            os.environ["MNN_GEMM_PAIR_VAR"] = mode[2:] if len(mode) > 1 else "0"
            t = timed(lambda: ops.gemm_tn(A, Bm, C, bias=bias))
            if rd > 0:
                res.setdefault((name, mode), []).append(t)
tot = {m: 0.0 for m in MODES}
for name, A, Bm, C, bias in bufs:
    M, K = A.shape
    N = Bm.shape[0]
    fl = 2.0 * M * N * K
    line = f"{name:12s} M={M} N={N:5d} K={K:5d}"
    for mode in MODES:
        ts = sorted(res[(name, mode)])
        med = ts[len(ts) // 2]
        tot[mode] += med
        line += f" | mode {mode}: {med*1e3:7.1f} us (min {ts[0]*1e3:7.1f}) {fl/med/1e9:6.0f} TF/s"
    print(line, flush=True)
print("sum of medians: " + ", ".join(f"mode {m}: {tot[m]:.3f} ms" for m in MODES))
