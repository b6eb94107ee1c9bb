import torch, sys, os
sys.path.insert(0, '.')
from multinn_amd import ops
dev = 'cuda'
def bench(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [("dWx1", 2048, 448, 262144), ("dWh1", 2048, 512, 262144), ("dWx2", 1024, 512, 262144), ("dWh2", 1024, 256, 262144), ("dWdense", 704, 256, 262144)]
tag = "128" if os.environ.get("MNN_GEMM_NO256") else "auto"
for name, M, N, K in shapes:
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16); Bm = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    out = []
    for sk in (4, 8, 16, 32, 64):
        out.append("sk%d %.0f" % (sk, bench(lambda: ops.gemm_tn(A, Bm, C, split_k=sk))))
    print(f"{tag} {name:8s}", "  ".join(out), flush=True)
