"""RnnRBM train step timing at the C3 per-track shape: [B=256, T=128, D=88], CD-10, eager launches."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import RnnRBM, AdamOptimizer
from multinn_amd import _lib

def main(B=256, T=128, D=88, k=10, steps=5):
    dev = "cuda:0"
    R = np.random.default_rng(23)
    x = (R.random((B, T + 1, D)) < 0.03).astype(np.float32)
    x[:, 0] = 0
    inp, tgt = torch.from_numpy(x[:, :-1]).to(dev), torch.from_numpy(x[:, 1:]).to(dev)
    g = RnnRBM(D, 256, [512, 256], keep_prob=0.9, k=k, precision="bf16", seed=23)
    opt = AdamOptimizer(0.01)
    def step():
        g.build(inp, tgt, None, True, "train")
        g.train(opt, 0.01)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    timing, _lib.TIMING = _lib.TIMING, None
    run = g.graphed_build_train(inp, tgt, opt, 0.01, warmup=1)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t0) / steps
    print(json.dumps({"graphed_ms_per_step": dtg * 1e3, "timesteps_per_s": B * T / dtg}))
    per = {k_: round(sum(a.elapsed_time(b) for a, b in v) / steps, 3) for k_, v in timing.items()}
    top = dict(sorted(per.items(), key=lambda kv: -kv[1])[:6])
    print(json.dumps({"B": B, "T": T, "D": D, "k": k, "ms_per_step": dt * 1e3, "timesteps_per_s": B * T / dt, "top_ms": top}))

if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
