import torch, time, sys
sys.path.insert(0, '.')
from multinn_amd import ops
dev = 'cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
# (M, N, K) of C[M,N] = A[M,K] . B[N,K]^T   -- the C2 train step's GEMMs
shapes = [("xproj1 fwd", 32768, 2048, 448), ("dense fwd", 32768, 704, 256), ("dense dgrad", 32768, 256, 704),
          ("dWx1", 2048, 448, 32768), ("dWh1", 2048, 512, 32768), ("dWx2", 1024, 512, 32768), ("dWh2", 1024, 256, 32768), ("dWdense", 704, 256, 32768)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); Bm = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    sk = max(1, min(1024 // max(-(-M // 128) * -(-N // 128), 1), K // 1024))
    t_mine = bench(lambda: ops.gemm_tn(A, Bm, C, split_k=sk))
    ref = (A.float() @ Bm.float().t())
    ops.gemm_tn(A, Bm, C, split_k=sk); torch.cuda.synchronize()
    err = float((C - ref).abs().max() / ref.abs().max())
    Bt = Bm.t()
    t_torch = bench(lambda: torch.mm(A, Bt))
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:6d}  mine {t_mine*1e3:7.1f} us {fl/t_mine/1e9:7.1f} TF/s (split_k {sk}, rel err {err:.1e}) | torch.mm {t_torch*1e3:7.1f} us {fl/t_torch/1e9:7.1f} TF/s")
