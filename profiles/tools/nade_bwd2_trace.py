"""Phase clocks of csrc/nade_bwd2.hip (wave 0 of workgroup 0; shader-clock cycles).  Build the traced library with
    python profiles/tools/build_trace_lib.py NB2_TRACE scratch/lib_nb2_trace.so
and run   MULTINN_HIP_LIB=scratch/lib_nb2_trace.so python profiles/tools/nade_bwd2_trace.py [rho]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from multinn_amd import ops, _lib

rho = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
N, D, Hn, tracks = 262144, 440, 256, 1
g = torch.Generator(device="cuda").manual_seed(1)
v = (torch.rand((tracks, N, D), device="cuda", generator=g) < rho).to(torch.uint8)
ld = (tracks * (Hn + D) + 63) // 64 * 64
bias = (torch.randn((N, ld), device="cuda", generator=g) * 0.5)[:, :tracks * (Hn + D)]
we = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
wd = torch.randn((tracks, D, Hn), device="cuda", generator=g) * 0.1
rw = torch.rand(N, device="cuda", generator=g) * 256.0
d0 = torch.zeros((N, ld), device="cuda")[:, :tracks * (Hn + D)]
af = torch.zeros((tracks, N, Hn), device="cuda")
ops.nade_logprob_fwd(v, bias, we, wd, tracks, D, Hn, rw, torch.zeros((tracks, N), device="cuda"), None, d0, af)
dwe, dwd = torch.zeros_like(we), torch.zeros_like(wd)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.nade_logprob_bwd(v, bias, we, wd, tracks, D, Hn, af, d0, dwe, dwd, f16_products=True)
e1.record()
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_longlong * 16)()
lib.mnn_nade_bwd2_trace.restype = C.c_int
assert lib.mnn_nade_bwd2_trace(buf) == 0
names = ["0 ballots + prefetch issue", "1 range search / prefix", "2 scatter operands", "3 wait B1", "4 S = AS.Wd + clear", "5 wait B2", "6 state machine", "7 wait B3",
         "8 chunk top (weights)", "9 C: dWd, dWe MFMAs", "10 flush atomics"]
tot = sum(buf[:11])
units = 4 * ((D + 31) // 32)
print(f"rho {rho}: launch {e0.elapsed_time(e1):.3f} ms; wave 0 of workgroup 0: {tot} cycles over {units} (sub-block, chunk) units = {tot / units:.0f} per unit")
for k, n in enumerate(names):
    print(f"  {n:32s} {buf[k]:10d}  {100.0 * buf[k] / tot:5.1f} %   {buf[k] / units:7.0f} per unit")
