"""Per-call timing of the bench-shape train step's plain GEMMs (TGT [1024,256,88,5]: N = 262 144 rows), with the C dtypes the step uses.
Run from the repository root on the GPU box:  python profiles/tools/gemm_probe_tgt.py [MULTINN_HIP_LIB=... for A/B builds]"""
import sys
import torch
sys.path.insert(0, '.')
from multinn_amd import ops
from multinn_amd.generators import LstmStack
dev = 'cuda'


def bench(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


Nr = 262144
# name, M, N, K, C dtype, bias, split-K (None = the step's rule for weight gradients)
shapes = [("xproj1 (bf16 C)", Nr, 2048, 448, torch.bfloat16, True, 1), ("xproj1 (f32 C)", Nr, 2048, 448, torch.float32, True, 1),
          ("xproj2 (bf16 C)", Nr, 1024, 512, torch.bfloat16, True, 1), ("xproj2 (f32 C)", Nr, 1024, 512, torch.float32, True, 1),
          ("dense fwd", Nr, 696, 256, torch.float32, True, 1), ("dense dgrad", Nr, 256, 704, torch.float32, False, 1),
          ("dX2 = dz2 . Wx2", Nr, 512, 1024, torch.float32, False, 1),
          ("dWx1", 2048, 448, Nr, torch.float32, False, None), ("dWh1", 2048, 512, Nr, torch.float32, False, None),
          ("dWx2", 1024, 512, Nr, torch.float32, False, None), ("dWh2", 1024, 256, Nr, torch.float32, False, None),
          ("dWdense", 256, 696, Nr, torch.float32, False, None)]
tot = 0.0
for name, M, N, K, cdt, hb, sk in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    Bm = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    ldc = (N + 63) // 64 * 64
    C = torch.empty(M, ldc, device=dev, dtype=cdt)[:, :N]
    bias = torch.randn(N, device=dev) if hb else None
    if sk is None:
        sk = LstmStack._split_k(M, N, K)
    t = bench(lambda: ops.gemm_tn(A, Bm, C, bias=bias, split_k=sk))
    fl = 2.0 * M * N * K
    byts = 2.0 * (M * K + N * K) + C.element_size() * M * N
    tot += t
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d} split {sk:3d}  {t*1e3:7.1f} us  {fl/t/1e9:7.1f} TF/s  {byts/t/1e6:7.1f} GB/s (operands once + C)")
print(f"sum {tot:.3f} ms")
