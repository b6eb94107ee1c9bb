#!/usr/bin/env python
"""bench.py -- throughput of one LSTM-NADE train step on synthetic 5-track piano-rolls.

    python bench.py --gpus N --steps K --warmup W [--workload c2|tgt] [--precision bf16|fp32]

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI); weak scaling: every
rank trains its own [B,T,88,5] batch, ONE all-reduce of the flat gradient per step.  Rank 0
prints ONE JSON line (contract in the task statement) with `roofline` for the dominant kernel and
`cpu_baseline` (the oracle's torch-CPU port of the reference formulation, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # BASELINE.json configs[1] and the north_star target shape
    "c2": dict(B=256, T=128, P=88, M=5, name="C2 joint LSTM-NADE [256,128,88,5]"),
    "tgt": dict(B=1024, T=256, P=88, M=5, name="TGT joint LSTM-NADE [1024,256,88,5]"),
    "tiny": dict(B=16, T=16, P=88, M=5, name="tiny [16,16,88,5] (plumbing check)"),
}
HN, UNITS = 256, [512, 256]          # default_params.yaml:11-12
PEAK_MFMA_BF16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def synth(B, T, P, M, seed, rho=0.03):
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((B, T, P, M)) < rho).astype(np.uint8)


def cpu_baseline(P, M, rho=0.03):
    """Bounded sample of the SAME workload on the host cores: the oracle's torch-CPU float32 port of the
    reference formulation (per-step LSTMBlockCell, per-visible NADE loop, autograd, clip + TF Adam)."""
    from oracle import generators as G, torch_ref as TR
    B, T = 8, 32
    D = P * M
    cores = min(16, os.cpu_count() or 1)      # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(cores)
    x = synth(B, T, P, M, 23, rho).astype(np.float32)
    inp, tgt = G.joint_inputs(x)
    Pm = TR.to_torch(G.init_rnn_nade(23, D, D, HN, UNITS, np.float32))
    opt = TR.TFAdam(TR.flat_params(Pm))
    du = [torch.tensor(a) for a in G.dropout_uniforms(23, B, T, UNITS)]
    xi, ti = torch.tensor(inp), torch.tensor(tgt)

    def step():
        loss, _, _ = TR.rnn_nade_loss(xi, ti, None, Pm, 0.9, du)
        loss.backward()
        opt.step()
    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 20):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return dict(value=B * T / dt, unit="timesteps/s", cores=cores, kind="port",
                sample=f"{n} train steps of [B={B},T={T},88,5] joint LSTM-NADE (oracle/torch_ref.py, float32, {cores} threads)")


def sampling_scan(gen, P, M, n=72, intro=32, steps=128, reps=3):
    """"sampling reported as generated timesteps/sec".  The scan of rnn_estimator.py:271-323 on the SAME generator the train
    step just used (joint LSTM-NADE): n intros of `intro` steps (default_config.yaml:43-51: 24 intros x 3), `steps` generated timesteps,
    one hipGraph replay per call.  Not part of the timed train-step region.  Two weight states: as trained by the timed steps (random-init
    b_dec ~ 0: every conditional near 0.5, half the draws are 1 -- the worst case for the scan, each 1 recomputes the 256 hidden
    sigmoids), and with the Dense bias of the b_dec block set to logit(0.03) (piano-roll-like draws)."""
    import math
    dev = "cuda"
    x = torch.from_numpy(synth(n, intro, P, M, 29).reshape(n, intro, P * M)).to(dev)
    res = {"unit": "generated timesteps/s", "n": n, "intro": intro, "steps": steps, "launch": "hipgraph-replay",
           "workload": f"joint LSTM-NADE sampling scan, {n} intros x {intro} steps -> {steps} generated steps"}

    def timed():
        out = gen.generate(x, steps)                      # captures on first use
        torch.cuda.synchronize()
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            out = gen.generate(x, steps)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, float(out.float().mean())

    t, dens = timed()
    res.update(value=n * steps / t, us_per_step=1e6 * t / steps, density=round(dens, 4))
    D, Hn = P * M, HN
    bias = gen.store["dense/bias"]
    saved = bias.clone()
    bias[Hn:Hn + D] = math.log(0.03 / 0.97)
    gen.store.step += 1                                   # weights changed: host-side packed copies are stale
    t, dens = timed()
    res["pianoroll_like"] = {"value": n * steps / t, "us_per_step": 1e6 * t / steps, "density": round(dens, 4)}
    bias.copy_(saved)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rho", type=float, default=0.03)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true", help="skip the sampling-scan measurement (rank 0, after the timed region)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on a 1-GPU box)")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearsal = os.environ.get("MULTINN_DP_REHEARSAL") == "1" and "RANK" in os.environ      # 1-rank RCCL run of the N>1 path
    if world > 1 or rehearsal:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    from multinn_amd import RnnNade, AdamOptimizer, _lib

    w = WORKLOADS[a.workload]
    B, T, P, M = w["B"], w["T"], w["P"], w["M"]
    D = P * M
    gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=a.precision, seed=23, device=dev)
    gen.row0 = rank * B                                  # RNG streams keyed by the GLOBAL sequence index
    x = torch.from_numpy(synth(B, T, P, M, 23 + rank, a.rho)).to(dev)
    opt = AdamOptimizer(0.01)

    def barrier():
        if world > 1 or rehearsal:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = not a.no_graph          # hipGraph replay of the step (N>1: forward+backward | eager all-reduce | clip+Adam)
    for _ in range(a.warmup):
        gen.train_step(x, None, opt)
    step_fn = lambda: gen.train_step(x, None, opt)
    if use_graph:
        try:
            step_fn = gen.graphed_train_step(x, opt, warmup=1)
        except Exception as e:      # keep the run alive: fall back to eager launches and say so in the JSON line
            print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
    if not use_graph:
        _lib.TIMING = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step_fn()
    barrier()
    dt = time.perf_counter() - t0
    if use_graph:
        # per-entry-point HIP-event breakdown: the same steps launched eagerly right after the timed region
        # (events cannot be recorded inside a graph replay)
        _lib.TIMING = {}
        for _ in range(a.steps):
            gen.train_step(x, None, opt)
        torch.cuda.synchronize()
    timing, _lib.TIMING = _lib.TIMING, None
    gen._stack.check()                  # a persistent launch that gave up on a bounded spin would have produced garbage: fail loudly
    if world > 1 or rehearsal:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    if rank != 0:
        if world > 1 or rehearsal:
            dist.barrier()              # rank 0 is still measuring the sampling scan: leave the process group together
            dist.destroy_process_group()
        return

    per_call = {k: (sum(e0.elapsed_time(e1) for e0, e1 in v) / a.steps, len(v) // a.steps) for k, v in timing.items()}
    total_ms = sum(v[0] for v in per_call.values())
    top = sorted(per_call.items(), key=lambda kv: -kv[1][0])
    print(f"# loss {float(loss):.4f}  step {dt / a.steps * 1e3:.2f} ms  sum(device) {total_ms:.2f} ms", file=sys.stderr)
    for k, (ms, n) in top:
        print(f"#   {k:32s} {ms:9.3f} ms/step  ({n} calls)", file=sys.stderr)

    N = B * T
    # dominant entry point and its roofline (algorithmic counts: DESIGN.md "Roofline accounting")
    dom, (dom_ms, dom_calls) = top[0]
    R1, R2 = UNITS
    peak_mfma = PEAK_MFMA_BF16_TFLOPS if a.precision == "bf16" else 157.3
    rec_flops = 2.0 * N * (R1 * 4 * R1 + R2 * 4 * R2)          # the T sequential [B,u]x[u,4u] products of both layers, one direction
    if dom in ("mnn_nade_logprob_bwd", "mnn_nade_logprob_fwd"):
        # NADE scan: VALU/transcendental work, almost no HBM; priced here against HBM with its algorithmic bytes
        byts = N * (D + 4 * (HN + D) * (2 if dom.endswith("bwd") else 1) + 4 * D)
        roof = dict(bound="hbm", achieved=byts / (dom_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", traffic=None, kernel=dom,
                    launches_per_step=1, avg_launch_us=dom_ms * 1e3,
                    note="VALU-bound scan (SURVEY 8d): the HBM fraction only shows HBM is not the limiter")
    elif dom in ("mnn_lstm2_persist_fwd", "mnn_lstm2_persist_bwd"):
        # ONE launch for the T-step recurrence of both layers; layer 2's input projection (its dgrad, backward) is folded in
        flops = rec_flops + 2.0 * N * R1 * 4 * R2
        roof = dict(bound="mfma", achieved=flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                    kernel="lstm2_persist_%s_kernel" % ("bwd" if dom.endswith("bwd") else "fwd"), launches_per_step=1,
                    avg_launch_us=dom_ms * 1e3, avg_timestep_us=dom_ms * 1e3 / T,
                    note="latency-bound chain of T in-kernel tile hand-offs (flag + 32-row tile through the fabric per timestep): "
                         "the number to watch is avg_timestep_us")
    elif dom in ("mnn_lstm_seq_fwd", "mnn_lstm_seq_bwd", "mnn_lstm2_seq_fwd", "mnn_lstm2_seq_bwd"):
        fused = dom.startswith("mnn_lstm2")
        launches = (T + 2) if fused else 2 * T            # fused: one three-stage launch per timestep for both layers (lag 2)
        roof = dict(bound="mfma", achieved=rec_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                    kernel=("lstm3_%s_step" if fused else "lstm_%s_step_v2") % ("bwd" if dom.endswith("bwd") else "fwd"),
                    launches_per_step=launches, avg_launch_us=dom_ms * 1e3 / launches,
                    note="latency-bound chain of T sequential launches: the number to watch is avg_launch_us")
    else:   # all plain GEMMs of the step: input projections, dense, their dgrad + wgrad, recurrent wgrad
        fwd = 2.0 * N * (D * 4 * R1 + R1 * 4 * R2 + R2 * (HN + D))
        flops = 3.0 * fwd - 2.0 * N * D * 4 * R1 + rec_flops
        roof = dict(bound="mfma", achieved=flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                    kernel="gemm_tn_glds_kernel", launches_per_step=dom_calls, avg_launch_us=dom_ms * 1e3 / dom_calls)
    roof["frac"] = roof["achieved"] / roof["peak"]
    # fabric-side bytes per launch of that kernel: rocprofv3 PMC passes recorded under profiles/ (FETCH_SIZE and WRITE_SIZE cannot
    # be collected from inside this process); only for the workload they were measured on
    try:
        pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_o_pmc_traffic.json")))
        if a.workload == "c2" and a.precision == "bf16":
            hit = [v for k, v in pmc["kernels"].items() if k.startswith(roof["kernel"])]
            if hit:
                roof["traffic"] = hit[0]["hbm_side_bytes_per_launch"]
                roof["traffic_unit"] = "bytes/launch"
                roof["traffic_source"] = "profiles/round1_o_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "piano-roll timesteps/sec (train step), 5-track LSTM-NADE", "value": world * B * T * a.steps / dt,
        "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32",
        "data": "synthetic",
        "config": {"workload": w["name"], "global_batch": world * B, "seq_len": T, "pitches": P, "tracks": M, "rho": a.rho,
                   "nade_hidden": HN, "lstm_units": UNITS, "keep_prob": 0.9, "optimizer": "TF-Adam lr 0.01 eps 1e-4 clip 5.0",
                   "parallelism": f"dp{world}"},
        "launch": "hipgraph-replay" if use_graph else "eager",
        "roofline": roof,
        "breakdown_ms": {k: round(v[0], 3) for k, v in top},
    }
    if rank == 0 and not a.no_sampling:
        out["sampling"] = sampling_scan(gen, P, M)
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(P, M, a.rho)
    print(json.dumps(out))
    if world > 1 or rehearsal:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
