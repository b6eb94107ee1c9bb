"""RnnMultiNADE train step at the C4 (composer) per-GPU shape: codes [B, T, E*M], E=84, M=5."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import RnnMultiNADE, AdamOptimizer
from multinn_amd import _lib

def main(B=128, T=128, E=84, M=5, steps=5):
    dev = "cuda:0"
    R = np.random.default_rng(23)
    seq = (R.random((B, T + 1, E * M)) < 0.2).astype(np.float32)
    x, y = torch.from_numpy(seq[:, :-1]).to(dev), torch.from_numpy(seq[:, 1:]).to(dev)
    g = RnnMultiNADE(E, 256, [512, 256], tracks=list("abcde")[:M], keep_prob=0.9, precision="bf16", seed=23)
    opt = AdamOptimizer(0.01)
    def step():
        g.build(x, y, None, True, "train")
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
    per = {k_: round(sum(a.elapsed_time(b) for a, b in v) / steps, 3) for k_, v in timing.items()}
    top = dict(sorted(per.items(), key=lambda kv: -kv[1])[:7])
    run = g.graphed_build_train(x, y, opt, 0.01, warmup=1)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t0) / steps
    print(json.dumps({"B": B, "T": T, "E": E, "M": M, "eager_ms": dt * 1e3, "graphed_ms": dtg * 1e3, "timesteps_per_s": B * T / dtg, "top_ms": top}))

if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
