"""Where a pass of nade_sample_chunk_kernel spends its time: per chunk of row 0, the wall clock (100 MHz) before / after the counted wait for the
chunk's LDS copies and after its passes, and the shader clock there.  Needs the SCH_TRACE build:
    python profiles/tools/build_trace_lib.py SCH_TRACE scratch/lib_sch_trace.so
    MULTINN_HIP_LIB=scratch/lib_sch_trace.so python profiles/tools/sample_chunk_trace.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import ops, _lib
DEV = "cuda:0"
N, D, Hn = 72, 440, 256
g = torch.Generator(device=DEV).manual_seed(5)
bias = torch.randn((N, Hn + D), device=DEV, generator=g) * 0.5
bias[:, Hn:] -= 3.5
we = torch.randn((1, D, Hn), device=DEV, generator=g) * 0.03
wd = torch.randn((1, D, Hn), device=DEV, generator=g) * 0.03
out = torch.zeros((N, D), device=DEV, dtype=torch.uint8)
for i in range(5):
    ops.nade_sample(bias, we, wd, 1, D, Hn, 1.0, 9, 77 + i, 3, out)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_longlong * (8 * 256))()
lib.mnn_sch_trace_read.restype = ctypes.c_int
assert lib.mnn_sch_trace_read(buf) == 0
G = 8 if os.environ.get("MNN_SAMPLE_G8") else 16          # (72 rows: every row has a CU to itself -> 16 visibles per pass unless told otherwise)
nch = (D + G - 1) // G
t = np.array(buf, dtype=np.int64).reshape(8, 256)[:, :nch]
ones = np.add.reduceat(out[0].cpu().numpy().astype(np.int64), np.arange(0, D, G))
wait = (t[1] - t[0]) * 10
work = (t[2] - t[1]) * 10
print("row 0: ones per chunk", ones.tolist())
print("wait ns per chunk   ", wait.tolist())
print("passes ns per chunk ", work.tolist())
print("first pass: weights out of LDS", ((t[4] - t[1]) * 10).tolist())
print("first pass: logits reduced    ", ((t[5] - t[4]) * 10).tolist())
print("first pass: draws decided     ", ((t[6] - t[5]) * 10).tolist())
print("rest (records, flips, restarts)", ((t[2] - t[6]) * 10).tolist())
print("between chunks (2 G copies issued)", ((t[0, 1:] - t[2, :-1]) * 10).tolist())
print(f"total {10 * (t[2, -1] - t[0, 0])} ns; waits {wait.sum()} ns; passes {work.sum()} ns; shader clock {(t[3, -1] - t[3, 0]) / (10 * (t[2, -1] - t[2, 0])):.3f} GHz")
