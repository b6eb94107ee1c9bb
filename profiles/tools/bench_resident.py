"""CU-resident vs row-parallel recurrence of one 256-unit layer: same inputs, outputs compared, both timed (HIP events on torch's stream).

    python profiles/tools/bench_resident.py [B] [T] [--keep 0.9] [--bf16]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multinn_amd import ops, _lib  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if len(args) > 0 else 1024
    T = int(args[1]) if len(args) > 1 else 256
    keep = float(sys.argv[sys.argv.index("--keep") + 1]) if "--keep" in sys.argv else 0.9
    dt = torch.bfloat16 if "--bf16" in sys.argv else torch.float16
    u, dev = 256, "cuda"
    g = torch.Generator(device="cpu").manual_seed(5)
    wh = (torch.randn((4 * u, u), generator=g) * 0.06).to(dev).to(dt)
    xproj = (torch.randn((T, B, 4 * u), generator=g) * 1.5).to(dev).to(dt)
    mask = (torch.rand((T, B, u), generator=g) < keep).to(torch.uint8).to(dev) if keep < 1.0 else None
    N = T * B

    def buffers():
        d = dict(gates=torch.zeros((T, B, 4 * u), device=dev, dtype=dt), c=torch.zeros((T, B, u), device=dev), h=torch.zeros((T, B, u), device=dev, dtype=dt),
                 hT=torch.zeros((u, N), device=dev, dtype=dt), yT=torch.zeros((u, N), device=dev, dtype=dt))
        d["y"] = torch.zeros((T, B, u), device=dev, dtype=dt) if mask is not None else None
        return d

    a, b = buffers(), buffers()
    da = ops.lstm2_fwd_layer(xproj, wh, None, None, a["gates"], a["c"], a["h"], a["hT"], a["y"], mask, yT=a["yT"], gates_dtype=dt, xproj_dtype=dt)
    db = ops.lstm2_fwd_layer(xproj, wh, None, None, b["gates"], b["c"], b["h"], b["hT"], b["y"], mask, yT=b["yT"], gates_dtype=dt, xproj_dtype=dt)
    out = {"B": B, "T": T, "keep": keep, "dtype": str(dt)}
    rp = ops.lstm_rowpar_ok(B, u)
    if rp:
        ws = ops.lstm_rowpar_workspace(T, B, u, dev)
        ops.lstm_rowpar_fwd(T, B, da, keep, ws)
    ops.lstm_resident_fwd(T, B, db, keep)
    torch.cuda.synchronize()
    if rp:
        ops.lstm_rowpar_check(ws)
        for k in a:
            if a[k] is None:
                continue
            x, y = a[k].float(), b[k].float()
            if k == "h" and mask is not None:
                x, y = x[-1], y[-1]
            out["maxdiff_" + k] = float((x - y).abs().max())
        out["rowpar_ms"] = timed(lambda: ops.lstm_rowpar_fwd(T, B, da, keep, ws))
    out["resident_ms"] = timed(lambda: ops.lstm_resident_fwd(T, B, db, keep))
    out["resident_us_per_step"] = out["resident_ms"] * 1e3 / T
    out["finite"] = bool(torch.isfinite(b["c"]).all())
    # ---- backward on the forward's saved tensors ----
    dh = (torch.randn((T, B, u), generator=g) * 0.01).to(dev)
    whp = wh.t().contiguous()
    kbl = N % 64 == 0 and B % 32 == 0 and "--plain" not in sys.argv

    def bbuf():
        return dict(dzc=torch.zeros((T, B, 4 * u), device=dev, dtype=dt),
                    dzT=torch.zeros((N // 32, 4 * u, 32), device=dev, dtype=dt) if kbl else torch.zeros((4 * u, N), device=dev, dtype=dt),
                    db=torch.zeros(4 * u, device=dev))
    ba, bb = bbuf(), bbuf()
    wsb = ops.lstm_seq_bwd_workspace(B, u, dev)
    src = a if rp else b
    ea = ops.lstm2_bwd_layer(dh, whp, src["gates"], src["c"], None, ba["dzc"], wsb, ba["dzT"], ba["db"], mask, gates_dtype=dt)
    eb = ops.lstm2_bwd_layer(dh, whp, src["gates"], src["c"], None, bb["dzc"], wsb, bb["dzT"], bb["db"], mask, gates_dtype=dt)
    if rp:
        ops.lstm_rowpar_bwd(T, B, ea, keep, ws)
    ops.lstm_resident_bwd(T, B, eb, keep)
    torch.cuda.synchronize()
    if rp:
        ops.lstm_rowpar_check(ws)
        for k in ba:
            x, y = ba[k].float(), bb[k].float()
            out["bwd_maxdiff_" + k] = float((x - y).abs().max())
            out["bwd_maxabs_" + k] = float(x.abs().max())
        out["rowpar_bwd_ms"] = timed(lambda: ops.lstm_rowpar_bwd(T, B, ea, keep, ws))
    out["resident_bwd_ms"] = timed(lambda: ops.lstm_resident_bwd(T, B, eb, keep))
    out["resident_bwd_us_per_step"] = out["resident_bwd_ms"] * 1e3 / T
    out["bwd_finite"] = bool(torch.isfinite(bb["dzc"].float()).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
