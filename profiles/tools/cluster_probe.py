"""512-unit LSTM recurrence: the cluster form (lstm_cluster.hip) against the row-parallel form (lstm_rowpar.hip) on the same inputs and buffers
layouts, in one process: results compared on small shapes (forward: h / y / c / gates / hT / yT; backward: dz, dzT, db), then both timed at the
bench shape (B = 1024, T = 256).      python profiles/tools/cluster_probe.py [fwd|both]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from multinn_amd import ops

DEV = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
u = 512


def make(T, B, keep, tdt, seed=3):
    g = torch.Generator(device="cuda").manual_seed(seed)
    wh_t = (torch.randn((4 * u, u), device=DEV, generator=g) * 0.04).to(tdt)
    xproj = (torch.randn((T, B, 4 * u), device=DEV, generator=g) * 1.2).to(tdt)
    mask = (torch.rand((T, B, u), device=DEV, generator=g) < keep).to(torch.uint8) if keep < 1.0 else None
    return wh_t, xproj, mask


def fwd(kind, T, B, keep, tdt, wh_t, xproj, mask, ws, save=True):
    N = T * B
    gates = torch.zeros((T, B, 4 * u), device=DEV, dtype=tdt) if save else None
    c = torch.zeros((T, B, u), device=DEV)
    h = torch.zeros((T, B, u), device=DEV, dtype=tdt)
    y = torch.zeros((T, B, u), device=DEV, dtype=tdt) if mask is not None else None
    hT = torch.zeros((u, N), device=DEV, dtype=tdt) if save else None
    yT = torch.zeros((u, N), device=DEV, dtype=tdt) if save else None
    L = ops.lstm2_fwd_layer(xproj, wh_t, None, None, gates, c, h, hT, y, mask, yT=yT, gates_dtype=tdt, xproj_dtype=tdt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    if kind == "cluster":
        ops.lstm_cluster_fwd(T, B, L, keep, ws)
    else:
        ops.lstm_rowpar_fwd(T, B, L, keep, ws)
    e1.record()
    torch.cuda.synchronize()
    ops.lstm_rowpar_check(ws)
    return e0.elapsed_time(e1), dict(gates=gates, c=c, h=h, y=y, hT=hT, yT=yT)


def bwd(kind, T, B, keep, tdt, wh_t, mask, saved, dh, ws, layout, rowmajor=True):
    N = T * B
    wh_p = wh_t.t().contiguous()
    kb = layout == "kblock"
    dzc = torch.zeros((T, B, 4 * u), device=DEV, dtype=tdt) if rowmajor else None      # (the first layer of a stack needs no row-major dz)
    dzT = torch.zeros((N // 32, 4 * u, 32), device=DEV, dtype=tdt) if kb else torch.zeros((4 * u, N), device=DEV, dtype=tdt)
    db = torch.zeros(4 * u, device=DEV)
    E = ops.lstm2_bwd_layer(dh, wh_p, saved["gates"], saved["c"], None, dzc, ops.lstm_seq_bwd_workspace(B, u, DEV), dzT, db, mask, gates_dtype=tdt)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    if kind == "cluster":
        ops.lstm_cluster_bwd(T, B, E, keep, ws)
    else:
        ops.lstm_rowpar_bwd(T, B, E, keep, ws)
    e1.record()
    torch.cuda.synchronize()
    ops.lstm_rowpar_check(ws)
    return e0.elapsed_time(e1), dict(dzc=dzc, dzT=dzT, db=db)


def diff(a, b):
    if a is None:
        return "-"
    return f"{float((a.double() - b.double()).abs().max()):.2e}"


for dt in (torch.float16, torch.bfloat16):
    for (T, B, keep, save) in [(5, 256, 0.9, True), (4, 512, 1.0, True), (6, 256, 0.9, False), (3, 1024, 0.9, True)]:
        assert ops.lstm_cluster_ok(B, u)
        ins = make(T, B, keep, dt)
        ws = ops.lstm_rowpar_workspace(T, B, u, DEV)
        _, r = fwd("rowpar", T, B, keep, dt, *ins, ws, save)
        _, k = fwd("cluster", T, B, keep, dt, *ins, ws, save)
        print(f"fwd {str(dt)[6:]} T={T} B={B} keep={keep} save={save}: max abs diff  " + "  ".join(f"{n} {diff(k[n], r[n])}" for n in ("h", "y", "c", "gates", "hT", "yT")), flush=True)

if what == "both":
    for dt in (torch.float16, torch.bfloat16):
        for (T, B, keep, layout) in [(5, 256, 0.9, "kblock"), (4, 512, 1.0, "plain"), (6, 1024, 0.9, "kblock")]:
            ins = make(T, B, keep, dt)
            ws = ops.lstm_rowpar_workspace(T, B, u, DEV)
            _, saved = fwd("rowpar", T, B, keep, dt, *ins, ws, True)
            dh = torch.randn((T, B, u), device=DEV) * 0.02
            _, r = bwd("rowpar", T, B, keep, dt, ins[0], ins[2], saved, dh, ws, layout)
            _, k = bwd("cluster", T, B, keep, dt, ins[0], ins[2], saved, dh, ws, layout)
            sc = float(r["dzc"].double().abs().max())
            print(f"bwd {str(dt)[6:]} T={T} B={B} keep={keep} {layout}: max |dz| {sc:.3e}  max abs diff dz {diff(k['dzc'], r['dzc'])}  dzT {diff(k['dzT'], r['dzT'])}  "
                  f"db {diff(k['db'], r['db'])} (max |db| {float(r['db'].abs().max()):.3e})  dzT == dz^T: "
                  f"{bool(torch.equal(k['dzT'].permute(0, 2, 1).reshape(T * B, 4 * u) if layout == 'kblock' else k['dzT'][:, :T * B].t(), k['dzc'].view(T * B, 4 * u)))}", flush=True)

T, B, keep, dt = 256, 1024, 0.9, torch.float16
ins = make(T, B, keep, dt)
ws = ops.lstm_rowpar_workspace(T, B, u, DEV)
tr, tc = [], []
for rd in range(5):
    tr.append(fwd("rowpar", T, B, keep, dt, *ins, ws)[0])
    t, k = fwd("cluster", T, B, keep, dt, *ins, ws)
    tc.append(t)
_, r = fwd("rowpar", T, B, keep, dt, *ins, ws)
print(f"fwd B=1024 T=256 fp16 keep=0.9: row-parallel {sorted(tr[1:])[1]:.3f} ms ({sorted(tr[1:])[1] / T * 1e3:.2f} us/step) | cluster {sorted(tc[1:])[1]:.3f} ms "
      f"({sorted(tc[1:])[1] / T * 1e3:.2f} us/step) | max abs diff h {diff(k['h'][-1], r['h'][-1])} y {diff(k['y'], r['y'])} c {diff(k['c'], r['c'])}", flush=True)
if what == "both":
    dh = torch.randn((T, B, u), device=DEV) * 0.02
    tr, tc = [], []
    for rd in range(5):
        tr.append(bwd("rowpar", T, B, keep, dt, ins[0], ins[2], r, dh, ws, "kblock")[0])
        t, kq = bwd("cluster", T, B, keep, dt, ins[0], ins[2], r, dh, ws, "kblock")
        tc.append(t)
    _, rq = bwd("rowpar", T, B, keep, dt, ins[0], ins[2], r, dh, ws, "kblock")
    print(f"bwd B=1024 T=256 fp16 keep=0.9: row-parallel {sorted(tr[1:])[1]:.3f} ms ({sorted(tr[1:])[1] / T * 1e3:.2f} us/step) | cluster {sorted(tc[1:])[1]:.3f} ms "
          f"({sorted(tc[1:])[1] / T * 1e3:.2f} us/step) | max |dz| {float(rq['dzc'].double().abs().max()):.3e} max abs diff dz {diff(kq['dzc'], rq['dzc'])} db {diff(kq['db'], rq['db'])}", flush=True)
    tr, tc = [], []
    for rd in range(5):
        tr.append(bwd("rowpar", T, B, keep, dt, ins[0], ins[2], r, dh, ws, "kblock", False)[0])
        tc.append(bwd("cluster", T, B, keep, dt, ins[0], ins[2], r, dh, ws, "kblock", False)[0])
    print(f"bwd B=1024 T=256 fp16 keep=0.9, no row-major dz (first layer): row-parallel {sorted(tr[1:])[1]:.3f} ms ({sorted(tr[1:])[1] / T * 1e3:.2f} us/step) | cluster "
          f"{sorted(tc[1:])[1]:.3f} ms ({sorted(tc[1:])[1] / T * 1e3:.2f} us/step)", flush=True)
