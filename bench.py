#!/usr/bin/env python
"""bench.py -- throughput of one LSTM-NADE train step on synthetic 5-track piano-rolls (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W [--workload tgt|c2|c1x5|tiny] [--precision bf16|fp32] [--rho 0.03]

Default workload = the north-star shape TGT [1024,256,88,5] (BASELINE.json `north_star`; C2 = configs[1] is `--workload c2`).
One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI); every rank trains its own [B,T,88,5] batch (weak
scaling), ONE all-reduce of the flat f32 gradient per step.  Started without WORLD_SIZE and with --gpus N > 1, this process only
LAUNCHES the N ranks (torch.distributed.run in a child process, before anything touches the GPU) and relays rank 0's line.

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      dominant entry point of the step (HIP events around every C-ABI call) against the MFMA / HBM peak, and
                roofline.step = T_min / T_measured with T_min from the DENSE algorithmic counts of SURVEY.md 8(d) (MFMA flops,
                NADE sigmoids against the sigmoid rate MEASURED here by mnn_probe_sigmoid, HBM bytes);
  rho05         the same step on rho = 0.5 input (the NADE kernels skip work where v = 0: this run keeps the number honest);
  fp32          the same step in the parity mode (precision="fp32": the mode that meets the 1e-4 gate);
  strong        (N > 1) the step with the GLOBAL batch fixed at B, B/N sequences per rank;
  sampling      generated timesteps/s of the sampling scan;
  cpu_baseline  the oracle's torch-CPU port of the reference formulation on the host cores (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # the north-star target shape, BASELINE.json configs[1], configs[0]'s batch / length with the benchmark's 5 tracks
    "tgt": dict(B=1024, T=256, P=88, M=5, name="TGT joint LSTM-NADE [1024,256,88,5]"),
    "c2": dict(B=256, T=128, P=88, M=5, name="C2 joint LSTM-NADE [256,128,88,5]"),
    "c1x5": dict(B=16, T=64, P=88, M=5, name="C1-sized joint LSTM-NADE [16,64,88,5]"),
    "tiny": dict(B=32, T=16, P=88, M=5, name="tiny [32,16,88,5] (plumbing check)"),
}
HN, UNITS = 256, [512, 256]          # default_params.yaml:11-12
PEAK_MFMA_BF16_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_MFMA_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="tgt", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--rho", type=float, default=0.03)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sampling", action="store_true", help="skip the sampling-scan measurement (rank 0, after the timed region)")
    ap.add_argument("--no-extras", action="store_true", help="skip the rho = 0.5, fp32 and strong-scaling legs")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N > 1 without RCCL)")
    ap.add_argument("--collective-only", action="store_true",
                    help="launcher / rendezvous check without a GPU: every rank only all-reduces a flat f32 buffer of the model's "
                         "gradient size (no kernels run; `value` is null)")
    return ap.parse_args(argv)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks as ONE child process tree
    (torch.distributed.run, loopback rendezvous) and relay rank 0's JSON line.  Runs before this process imports torch.cuda or
    touches the GPU: a process that has initialised the GPU must never exec / be replaced (task environment rule)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL peer access)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{"):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


def synth(B, T, P, M, seed, rho=0.03):
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((B, T, P, M)) < rho).astype(np.uint8)


def cpu_baseline(P, M, rho=0.03):
    """Bounded sample of the SAME workload on the host cores: the oracle's torch-CPU float32 port of the reference formulation
    (per-step LSTMBlockCell, per-visible NADE loop keeping every [N,Hn] sigmoid for autograd, clip + TF Adam) at C1's batch and
    length with the benchmark's five tracks ([16,64,88,5]: 1024 rows per step; TGT needs 118 GB in this formulation)."""
    import numpy as np
    import torch
    from oracle import generators as G, torch_ref as TR
    B, T = 16, 64
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
    while n < 2 or (time.perf_counter() - t0 < 15.0 and n < 20):
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
    import torch
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


def sigmoid_peak(dev):
    """The device's MEASURED sigmoid throughput: mnn_probe_sigmoid = 8 independent chains per thread of the NADE
    kernels' own sigmoid (v_mul, v_exp_f32, v_add, v_rcp_f32), 8 waves per SIMD on every CU, timed with HIP events."""
    import torch
    from multinn_amd import _lib
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks, iters = cus * 8, 4096
    out = torch.empty(blocks * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    best = float("inf")
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.mnn_probe_sigmoid(st, blocks, iters, out.data_ptr()), "mnn_probe_sigmoid")
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    n = blocks * 256 * 8 * iters
    return dict(sigmoids_per_s=n / best, launch_ms=best * 1e3, sigmoids=n, cus=cus,
                source="mnn_probe_sigmoid timed with HIP events in this run (8 chains/thread, 8 waves/SIMD)")


def step_roofline(N, D, precision, t_measured_s, sig_peak):
    """T_min = sum over phases of max(dense FLOP / MFMA peak, transcendentals / measured peak, bytes / HBM peak),
    always from the DENSE algorithmic counts (the kernels' exact input-sparsity shortcuts are NOT credited).
    Phase 1 (LSTM + Dense, forward and backward): 3 x 2 x [(D+R1) 4R1 + (R1+R2) 4R2 + R2 (D+Hn)] FLOP per row on MFMA; bytes:
    uint8 input, saved LSTM activations written + read, weights / gradient / Adam slots.
    Phase 2 (NADE scan, forward and backward): D Hn + D sigmoids forward, D Hn sigmoids of recomputed hidden states backward,
    2 x 2 x 2 D Hn FLOP on the vector ALUs (not priced: transcendental-bound); bytes: [b_enc | b_dec] written + read twice
    (forward, backward) and its gradient written + read."""
    R1, R2 = UNITS
    fwd_flop = 2.0 * ((D + R1) * 4 * R1 + (R1 + R2) * 4 * R2 + R2 * (D + HN))
    flops = 3.0 * fwd_flop * N
    peak_mfma = (PEAK_MFMA_F32_TFLOPS if precision == "fp32" else PEAK_MFMA_BF16_TFLOPS) * 1e12      # f16 and bf16 MFMA: the same dense rate
    act = 4 if precision == "fp32" else 2
    bytes_dense = N * D + 2.0 * N * (4 * R1 + 4 * R2) * 4 + 2.0 * N * (2 * R1 + 2 * R2) * act + 7 * 4 * 3143352
    sig = N * (2.0 * D * HN + D)
    bytes_nade = N * (3 * 4 * (HN + D) + 2 * 4 * (HN + D) + D)
    t_mfma, t_hbm1 = flops / peak_mfma, bytes_dense / (PEAK_HBM_GBS * 1e9)
    t_sig, t_hbm2 = sig / sig_peak, bytes_nade / (PEAK_HBM_GBS * 1e9)
    t_min = max(t_mfma, t_hbm1) + max(t_sig, t_hbm2)
    return dict(t_min_ms=t_min * 1e3, t_measured_ms=t_measured_s * 1e3, frac=t_min / t_measured_s,
                phases={"lstm_dense": {"mfma_ms": t_mfma * 1e3, "hbm_ms": t_hbm1 * 1e3, "flop": flops, "bytes": bytes_dense},
                        "nade_scan": {"sigmoid_ms": t_sig * 1e3, "hbm_ms": t_hbm2 * 1e3, "sigmoids": sig, "bytes": bytes_nade}},
                accounting="dense algorithmic counts (SURVEY.md 8d); input-sparsity shortcuts not credited")


def collective_only(a):
    """Launcher / rendezvous rehearsal without a GPU (tests/test_bench_launch.py): N ranks, `gloo`, the flat gradient buffer of the
    joint LSTM-NADE model all-reduced K times with the same barrier + max-over-ranks timing protocol as the real run."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        dist.init_process_group(a.backend if a.backend != "nccl" else "gloo")
    grad = torch.full((3143352,), float(rank + 1))
    for _ in range(a.warmup):
        if world > 1:
            dist.all_reduce(grad.clone())
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g = grad.clone()
        if world > 1:
            dist.all_reduce(g)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ok = abs(float(g[0]) - world * (world + 1) / 2) < 1e-6
    if rank == 0:
        w = WORKLOADS[a.workload]
        print(json.dumps({"metric": "piano-roll timesteps/sec (train step), 5-track LSTM-NADE", "value": None, "unit": "timesteps/s",
                          "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                          "config": {"workload": w["name"], "parallelism": f"dp{world}"}, "collective_only": True,
                          "allreduce_ok": ok, "allreduce_bytes": grad.numel() * 4}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    a = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a.gpus, sys.argv[1:] if argv is None else argv)
    if a.collective_only:
        return collective_only(a)

    import numpy as np
    import torch
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearsal = os.environ.get("MULTINN_DP_REHEARSAL") == "1" and "RANK" in os.environ      # 1-rank RCCL run of the N>1 path
    multi = world > 1 or rehearsal
    if multi:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
    from multinn_amd import RnnNade, AdamOptimizer, _lib

    w = WORKLOADS[a.workload]
    B, T, P, M = w["B"], w["T"], w["P"], w["M"]
    D = P * M
    N = B * T

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(precision, rho, batch, steps, warmup, keep=False):
        """`steps` optimiser steps of a fresh generator on a resident synthetic batch: W eager warm-up steps, capture, K replays
        bracketed by barrier + synchronize, MAX over ranks.  Returns (seconds per step, loss, generator or None, launch mode)."""
        gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23, device=dev)
        gen.row0 = rank * batch                               # RNG streams keyed by the GLOBAL sequence index
        x = torch.from_numpy(synth(batch, T, P, M, 23 + rank, rho)).to(dev)
        opt = AdamOptimizer(0.01)
        for _ in range(warmup):
            gen.train_step(x, None, opt)
        step_fn = lambda: gen.train_step(x, None, opt)
        graph = not a.no_graph          # hipGraph replay of the step (N>1: forward+backward | eager all-reduce | clip+Adam)
        if graph:
            try:
                step_fn = gen.graphed_train_step(x, opt, warmup=1)
            except Exception as e:      # keep the run alive: fall back to eager launches and say so in the JSON line
                print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph = False
                torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step_fn()
        barrier()
        dt = time.perf_counter() - t0
        gen._stack.check()              # a persistent launch that gave up on a bounded spin would have produced garbage: fail loudly
        if multi:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        res = (dt / steps, float(loss), (gen, x, opt) if keep else None, "hipgraph-replay" if graph else "eager")
        if not keep:
            del gen, x, step_fn
            torch.cuda.empty_cache()
        return res

    # ---- the timed region: K steps of the named workload, weak scaling -------------------------------------------------------
    sec, loss, kept, launch = timed_steps(a.precision, a.rho, B, a.steps, a.warmup, keep=True)
    gen, x, opt = kept
    # per-entry-point HIP-event breakdown: the same steps launched eagerly right after the timed region (events cannot be recorded
    # inside a graph replay); every C-ABI call is bracketed by events on the stream it launches on
    _lib.TIMING = {}
    nb = min(a.steps, 5)
    for _ in range(nb):
        gen.train_step(x, None, opt)
    torch.cuda.synchronize()
    timing, _lib.TIMING = _lib.TIMING, None

    extras = {}
    if not a.no_extras and multi and world > 1 and B % world == 0 and (B // world) % 32 == 0:
        s2, _, _, _ = timed_steps(a.precision, a.rho, B // world, a.steps, a.warmup)
        extras["strong"] = {"global_batch": B, "per_gpu_batch": B // world, "ms_per_step": s2 * 1e3, "value": B * T / s2, "unit": "timesteps/s"}

    if rank != 0:
        if multi:
            dist.barrier()              # rank 0 is still measuring the single-rank legs: leave the process group together
            dist.destroy_process_group()
        return 0

    per_call = {k: (sum(e0.elapsed_time(e1) for e0, e1 in v) / nb, len(v) // nb) for k, v in timing.items()}
    total_ms = sum(v[0] for v in per_call.values())
    top = sorted(per_call.items(), key=lambda kv: -kv[1][0])
    print(f"# loss {loss:.4f}  step {sec * 1e3:.2f} ms  sum(device, eager) {total_ms:.2f} ms", file=sys.stderr)
    for k, (ms, n) in top:
        print(f"#   {k:32s} {ms:9.3f} ms/step  ({n} calls)", file=sys.stderr)

    # ---- roofline per entry point (algorithmic counts: DESIGN.md "Roofline accounting"); `roofline` = the one with the largest share ----
    R1, R2 = UNITS
    peak_mfma = PEAK_MFMA_F32_TFLOPS if a.precision == "fp32" else PEAK_MFMA_BF16_TFLOPS
    rec_flops = 2.0 * N * (R1 * 4 * R1 + R2 * 4 * R2)          # the T sequential [B,u]x[u,4u] products of both layers, one direction

    def entry_roofline(dom, dom_ms, dom_calls):
        if dom.startswith("mnn_nade_logprob"):
            bwd = dom.endswith("bwd")
            byts = N * (D + 4 * (HN + D) * (2 if bwd else 1) + 4 * (HN + D if bwd else D))
            roof = dict(bound="hbm", achieved=byts / (dom_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", traffic=None,
                        kernel="nade_bwd_kernel" if bwd else ("nade_fwd_mfma_kernel" if a.precision == "bf16" else "nade_fwd_kernel"),
                        entry_point=dom, launches_per_step=1, avg_launch_us=dom_ms * 1e3, algorithmic_bytes_per_launch=byts,
                        note="transcendental / VALU-bound scan (SURVEY 8d): HBM is the contract's bound for a non-MFMA kernel; its "
                             "sigmoid-rate fraction is in roofline.step.phases")
        elif dom in ("mnn_lstm2_persist_fwd", "mnn_lstm2_persist_bwd"):
            # ONE launch for the T-step recurrence of both layers; layer 2's input projection (its dgrad, backward) is folded in
            flops = rec_flops + 2.0 * N * R1 * 4 * R2
            roof = dict(bound="mfma", achieved=flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="lstm2_persist_%s_kernel" % ("bwd" if dom.endswith("bwd") else "fwd"), launches_per_step=1,
                        avg_launch_us=dom_ms * 1e3, avg_timestep_us=dom_ms * 1e3 / T, algorithmic_flop_per_launch=flops,
                        note="latency-bound chain of T in-kernel tile hand-offs (flag + row tile through the fabric per timestep): "
                             "the number to watch is avg_timestep_us")
        elif dom in ("mnn_lstm_rowpar_fwd", "mnn_lstm_rowpar_bwd"):
            # one launch per LAYER: every wave carries one 32-row tile through all T steps against a 32-unit weight tile resident in LDS
            roof = dict(bound="mfma", achieved=rec_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="lstm_rowpar_%s" % ("bwd" if dom.endswith("bwd") else "fwd"), entry_point=dom, launches_per_step=dom_calls,
                        avg_launch_us=dom_ms * 1e3 / dom_calls, avg_timestep_us=dom_ms * 1e3 / dom_calls / T,
                        algorithmic_flop_per_launch=rec_flops / dom_calls,
                        note="latency-bound chain: per timestep every row tile exchanges its 32-row h (backward: dz) slice with the other unit "
                             "tiles through L2 (flag per wave); the number to watch is avg_timestep_us")
        elif dom in ("mnn_lstm_seq_fwd", "mnn_lstm_seq_bwd", "mnn_lstm2_seq_fwd", "mnn_lstm2_seq_bwd"):
            fused = dom.startswith("mnn_lstm2")
            launches = (T + 2) if fused else 2 * T            # fused: one three-stage launch per timestep for both layers (lag 2)
            roof = dict(bound="mfma", achieved=rec_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel=("lstm3_%s_step" if fused else "lstm_%s_step_v2") % ("bwd" if dom.endswith("bwd") else "fwd"),
                        launches_per_step=launches, avg_launch_us=dom_ms * 1e3 / launches, algorithmic_flop_per_launch=rec_flops / launches,
                        note="latency-bound chain of T sequential launches: the number to watch is avg_launch_us")
        else:   # all plain GEMMs of the step: input projections, dense, their dgrad + wgrad, recurrent wgrad
            fwd = 2.0 * N * (D * 4 * R1 + R1 * 4 * R2 + R2 * (HN + D))
            flops = 3.0 * fwd - 2.0 * N * D * 4 * R1 + rec_flops
            roof = dict(bound="mfma", achieved=flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="gemm_tn_glds", launches_per_step=dom_calls, avg_launch_us=dom_ms * 1e3 / dom_calls,
                        algorithmic_flop_per_launch=flops / dom_calls)
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof.setdefault("entry_point", dom)
        return roof

    dom, (dom_ms, dom_calls) = top[0]
    roof = entry_roofline(dom, dom_ms, dom_calls)
    roof["others"] = [entry_roofline(k, ms, n) for k, (ms, n) in top[1:5] if ms > 0.05 * total_ms]
    # fabric-side bytes per launch of each kernel: rocprofv3 PMC passes recorded under profiles/ (FETCH_SIZE and WRITE_SIZE cannot be
    # collected from inside this process); only for the workload they were measured on
    def attach_traffic(rf):
        for fn in ("round2_%s_pmc_traffic.json" % a.workload, "round1_o_pmc_traffic.json" if a.workload == "c2" else ""):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                if a.precision == "bf16":
                    hit = [v for k, v in pmc["kernels"].items() if k.startswith(rf["kernel"])]
                    if hit:
                        rf["traffic"] = sum(h["hbm_side_bytes_per_launch"] * h["calls"] for h in hit) / sum(h["calls"] for h in hit)
                        rf["traffic_unit"] = "bytes/launch"
                        rf["traffic_source"] = f"profiles/{fn} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)"
                        return
            except (OSError, ValueError, KeyError):
                continue

    roof["timing"] = "HIP events around every C-ABI call of %d eager steps run right after the timed replays" % nb
    attach_traffic(roof)
    for rf in roof["others"]:
        attach_traffic(rf)

    sig = sigmoid_peak(dev)
    roof["step"] = step_roofline(N, D, a.precision, sec, sig["sigmoids_per_s"])
    roof["step"]["sigmoid_peak"] = sig
    roof["step"]["rho"] = a.rho

    out = {
        "metric": "piano-roll timesteps/sec (train step), 5-track LSTM-NADE", "value": world * B * T / sec,
        "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"fp16": "fp16", "bf16": "bf16", "fp32": "f32"}[a.precision],
        "data": "synthetic",
        "config": {"workload": w["name"], "global_batch": world * B, "per_gpu_batch": B, "seq_len": T, "pitches": P, "tracks": M, "rho": a.rho,
                   "nade_hidden": HN, "lstm_units": UNITS, "keep_prob": 0.9, "optimizer": "TF-Adam lr 0.01 eps 1e-4 clip 5.0",
                   "parallelism": f"dp{world}"},
        "launch": launch,
        "roofline": roof,
        "breakdown_ms": {k: round(v[0], 3) for k, v in top},
    }
    out.update(extras)
    if not a.no_sampling:
        out["sampling"] = sampling_scan(gen, P, M)
    del gen, x, opt, kept
    torch.cuda.empty_cache()
    if world == 1 and not rehearsal and not a.no_extras:
        # dense stress input: the NADE kernels' exact sparsity shortcuts vanish at rho = 0.5
        s5, l5, _, _ = timed_steps(a.precision, 0.5, B, max(3, a.steps // 2), 2)
        r5 = step_roofline(N, D, a.precision, s5, sig["sigmoids_per_s"])
        out["rho05"] = {"rho": 0.5, "ms_per_step": s5 * 1e3, "value": B * T / s5, "unit": "timesteps/s", "loss": l5,
                        "roofline_step_frac": r5["frac"], "t_min_ms": r5["t_min_ms"]}
        if a.precision == "bf16":
            sf, lf, _, lm = timed_steps("fp32", a.rho, B, 3, 1)
            rf = step_roofline(N, D, "fp32", sf, sig["sigmoids_per_s"])
            out["fp32"] = {"ms_per_step": sf * 1e3, "value": B * T / sf, "unit": "timesteps/s", "steps": 3, "loss": lf, "launch": lm,
                           "roofline_step_frac": rf["frac"], "t_min_ms": rf["t_min_ms"],
                           "note": "precision='fp32': v_mfma_f32_32x32x2_f32 GEMMs, launch-per-timestep recurrence, f32 NADE kernels; "
                                   "the mode the 1e-4 parity tests run in"}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(P, M, a.rho)
    print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
