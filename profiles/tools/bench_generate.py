"""Sampling-scan timing (A10/A11): intro pass + num_steps x {NADE sample, LSTM step, Dense}."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import RnnNade

def main(n=72, Ti=32, steps=128, reps=3):
    dev = "cuda:0"
    R = np.random.default_rng(23)
    x = torch.from_numpy((R.random((n, Ti, 440)) < 0.03).astype(np.uint8)).to(dev)
    g = RnnNade(440, 256, [512, 256], keep_prob=0.9, precision="bf16", seed=23)
    out = g.generate(x, 4)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = g.generate(x, steps)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(json.dumps({"n": n, "intro": Ti, "steps": steps, "s": t, "us_per_step": 1e6 * t / steps, "generated_timesteps_per_s": n * steps / t,
                      "density": float(out.float().mean())}))

if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
