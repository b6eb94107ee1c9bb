"""C5 sampling scan timing (A19): M per-track LSTM-NADE generators + feedback LSTM, n intros, T generated steps."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from multinn_amd import RnnNade, FeedbackRnn, FeedbackRnnSampler

def main(n=72, Ti=32, steps=128, reps=3):
    dev = "cuda:0"
    P, M, Hn, F = 88, 5, 256, 128
    R = np.random.default_rng(23)
    x = torch.from_numpy((R.random((n, Ti, P, M)) < 0.03).astype(np.uint8)).to(dev)
    fb = FeedbackRnn(P * M, [256, F], precision="bf16", seed=40)
    gens = [RnnNade(P, Hn, [256, 256], precision="bf16", seed=50 + i) for i in range(M)]
    for g in gens:
        g._materialize(P + F)
    smp = FeedbackRnnSampler(gens, fb)
    out = smp.generate(x, steps)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = smp.generate(x, steps)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(json.dumps({"n": n, "intro": Ti, "steps": steps, "us_per_step": 1e6 * t / steps, "generated_timesteps_per_s": n * steps / t,
                      "density": float(out.float().mean())}))

if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
