"""Stage clocks of lstm_step_det_kernel (workgroup 0, its four waves): start -> inputs staged -> barrier -> chain done -> partial sums met -> end.
Needs the DS_TRACE build:
    python profiles/tools/build_trace_lib.py DS_TRACE scratch/lib_ds_trace.so
    MULTINN_HIP_LIB=scratch/lib_ds_trace.so python profiles/tools/det_step_trace.py"""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import ops, _lib
DEV = "cuda:0"
B = 72
lib = _lib.load()
lib.mnn_ds_trace_read.restype = ctypes.c_int
for n_x, u in ((440, 512), (512, 256)):
    g = torch.Generator(device=DEV).manual_seed(3)
    x = (torch.rand((B, n_x), device=DEV, generator=g) < 0.03).to(torch.uint8) if n_x == 440 else torch.randn((B, n_x), device=DEV, generator=g)
    W = torch.randn((n_x + u, 4 * u), device=DEV, generator=g) * 0.05
    b = torch.zeros(4 * u, device=DEV)
    hp, cp = torch.randn((B, u), device=DEV, generator=g) * 0.3, torch.randn((B, u), device=DEV, generator=g) * 0.3
    ho, co = torch.empty_like(hp), torch.empty_like(cp)
    junk = torch.empty(64 << 20, device=DEV)                       # flush the L2s between calls, as the other kernels of a scan step do
    job = dict(x=x, n_x=n_x, h_prev=hp, c_prev=cp, W=W, bias=b, c_out=co, h_out=ho)
    if "--packed" in sys.argv:
        job["Wp"] = ops.det_lstm_pack(W, u)
    for hot in (True, False):
        for _ in range(3):
            if not hot: junk.zero_()
            ops.lstm_step_det([job])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.lstm_step_det([job]); e1.record(); torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 32)()
        assert lib.mnn_ds_trace_read(buf) == 0
        t = np.array(buf, dtype=np.int64).reshape(8, 4)[:6] * 10
        t = t - t[0].min()
        print(f"n_x={n_x} units={u} K={n_x + u} {'weights hot' if hot else 'L2 flushed'}: kernel {1e3 * e0.elapsed_time(e1):.1f} us (events)")
        for k, name in enumerate(("start", "inputs staged", "barrier passed", "chain done", "partial sums met", "end")):
            print(f"   {name:18s} ns per wave {t[k].tolist()}")
