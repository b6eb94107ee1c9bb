#!/usr/bin/env python
"""bench.py -- throughput of one LSTM-NADE train step on synthetic 5-track piano-rolls (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W [--workload tgt|c2|c1x5|tiny|c3|c4] [--precision fp16|bf16|fp32] [--rho 0.03]

Default workload = the north-star shape TGT [1024,256,88,5] (BASELINE.json `north_star`; C2 = configs[1] is `--workload c2`), default
precision = fp16: the mode for which tests/test_gpu_realdims.py asserts BASELINE.json's 1e-4 on loss, per-row NLL and conditionals at
D = 440 / Hn = 256 / [512,256] with the persistent kernels ON.  One process per GPU (torch.distributed, backend "nccl" = RCCL over
xGMI); every rank trains its own [B,T,88,5] batch (weak scaling), ONE all-reduce of the flat f32 gradient per step.  Started without
WORLD_SIZE and with --gpus N > 1, this process only LAUNCHES the N ranks (torch.distributed.run in a child process, before anything touches
the GPU) and relays rank 0's line.

Rank 0 prints ONE JSON line: the contract fields plus
  roofline      dominant entry point of the step (HIP events around every C-ABI call) against the MFMA / HBM peak, its counters from the
                committed rocprofv3 passes (mfma_busy, hbm_gbs, traffic), and roofline.step: per PHASE (LSTM + Dense | NADE scan) the measured
                time, T_min and frac under two accountings -- `dense` = SURVEY.md 8(d)'s counts (every sigmoid of the reference formulation)
                and `executed` = the FMAs / sigmoids / MFMA flops the kernels really issue at this batch's density (counted on the device).
                roofline.step.frac uses `executed`; a phase fraction above 1 sets "invalid": true (an accounting the kernels do not execute);
  rho05         the same step on rho = 0.5 input (the NADE kernels skip work where v = 0: this run keeps the number honest);
  ragged        the step on lengths ~ U{T/2..T} (seed 24): Dense + NADE on the valid rows only (device-side compaction), captured like the full step;
  bf16 / fp32   the same step in the other two modes, with their loss against the headline mode's on the same batch;
  strong        (N > 1) the step with the GLOBAL batch fixed at B, B/N sequences per rank;
  strong_proxy  (N = 1) the step at B/8 sequences: the per-GPU share of the global batch at 8 GPUs (8 x its rate = the strong leg's ceiling);
  sampling      generated timesteps/s of the sampling scan;
  cpu_baseline  the oracle's torch-CPU port of the reference formulation on the host cores (N = 1 only): median of 3 steps at C2 [256,128,88,5],
                preparation (batch, dropout uniforms, conversions) outside the timed region and reported separately.
`--workload c3|c4` (BASELINE configs[2] / [3]: jamming 5 x LSTM-RBM CD-10, composer DBNEncoder -> LSTM-MultiNADE) time the mode classes'
captured train step at the reference's layer widths and print the same contract fields.
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
    "tgt8": dict(B=128, T=256, P=88, M=5, name="TGT / 8 joint LSTM-NADE [128,256,88,5] (the per-GPU share of the strong-scaling leg)"),
    "c2": dict(B=256, T=128, P=88, M=5, name="C2 joint LSTM-NADE [256,128,88,5]"),
    "c1x5": dict(B=16, T=64, P=88, M=5, name="C1-sized joint LSTM-NADE [16,64,88,5]"),
    "tiny": dict(B=32, T=16, P=88, M=5, name="tiny [32,16,88,5] (plumbing check)"),
    # the other BASELINE configurations, through multinn_amd.modes (B, T as SURVEY 8(d) assumes them)
    "c3": dict(B=256, T=128, P=88, M=5, name="C3 jamming: 5 x LSTM-RBM(88, 256, [512,256], CD-10) [256,128,88,5]", mode="jamming"),
    "c4": dict(B=1024, T=128, P=88, M=5, name="C4 composer: DBNEncoder[168,84] x 5 -> LSTM-MultiNADE [1024,128,88,5] (per-GPU batch)", mode="composer"),
}
HN, UNITS = 256, [512, 256]          # default_params.yaml:11-12
PEAK_MFMA_16_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 / f16 MFMA
PEAK_MFMA_F32_TFLOPS = 157.3
PEAK_VALU_F32_TFLOPS = 157.3         # f32 vector FMA rate (2 flop per FMA)
PEAK_HBM_GBS = 8000.0
N_PARAMS = 3143352                   # joint LSTM-NADE, D = 440
DTYPE = {"fp16": "fp16", "bf16": "bf16", "fp32": "f32"}


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
    ap.add_argument("--no-extras", action="store_true", help="skip the rho = 0.5, ragged, bf16, fp32 and strong-scaling legs")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N > 1 without RCCL)")
    ap.add_argument("--collective-only", action="store_true",
                    help="launcher / rendezvous check without a GPU: every rank only all-reduces a flat f32 buffer of the model's "
                         "gradient size (no kernels run; `value` is null)")
    return ap.parse_args(argv)


def launch_ranks(n, argv, need_devices=True):
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks as ONE child process tree
    (torch.distributed.run, loopback rendezvous) and relay rank 0's JSON line.  Runs before this process touches the GPU: a process that
    has initialised the GPU must never exec / be replaced (task environment rule); counting devices does not initialise it."""
    if need_devices:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"# bench.py --gpus {n}: only {have} device(s) visible; refusing to start ranks that would share a GPU", file=sys.stderr)
            return 2
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


def ragged_lengths(B, T, seed=24):
    """lengths ~ U{T/2..T}, seed 24."""
    import numpy as np
    return np.random.Generator(np.random.PCG64(seed)).integers(T // 2, T + 1, size=B).astype(np.int32)


def cpu_baseline(P, M, rho=0.03, reps=3):
    """The SAME train step on the host cores: the oracle's torch-CPU float32 port of the reference formulation (per-step LSTMBlockCell,
    per-visible NADE loop keeping every [rows, Hn] sigmoid for autograd, clip + TF Adam) at C2 [256,128,88,5] (BASELINE.md names C1 / C2 for
    the CPU path; TGT needs 118 GB in this formulation).  The NADE part runs in row chunks with its backward inside the chunk (the reference
    keeps all [N, D, Hn] activations: 14.8 GB at C2), which changes memory, not arithmetic.

    Only the step is timed: the synthetic batch, its input / target slices, the dropout uniforms (25 M NumPy-Philox draws) and every tensor
    conversion are PREPARED first (reported as `prepare_seconds`), one full-shape step runs untimed (allocator, thread pool, autograd graph
    caches), then `reps` steps are timed one by one and the MEDIAN is reported (`seconds`; all of them in `seconds_each`)."""
    import numpy as np
    import torch
    from oracle import generators as G, torch_ref as TR
    B, T = 256, 128
    D = P * M
    cores = min(16, os.cpu_count() or 1)      # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(cores)
    Pm = TR.to_torch(G.init_rnn_nade(23, D, D, HN, UNITS, np.float32))
    opt = TR.TFAdam(TR.flat_params(Pm))

    def prepare(B, T):
        x = synth(B, T, P, M, 23, rho).astype(np.float32)
        inp, tgt = G.joint_inputs(x)
        du = [torch.tensor(a) for a in G.dropout_uniforms(23, B, T, UNITS)]
        return torch.tensor(inp), torch.tensor(tgt).reshape(B * T, D), du

    def step(xi, ti, du, chunk=2048):
        N = ti.shape[0]
        y, _ = TR.lstm_seq(xi, Pm['lstm'], 0.9, du)
        out = y.reshape(N, -1) @ Pm['fc_k'] + Pm['fc_b']
        d_out = torch.zeros_like(out)
        loss = 0.0
        for s in range(0, N, chunk):            # forward + backward of the NADE scan per row chunk (weight gradients accumulate in .grad)
            oc = out[s:s + chunk].detach().requires_grad_(True)
            nll, _ = TR.nade_log_prob(ti[s:s + chunk], oc[:, :HN], oc[:, HN:HN + D], Pm['w_enc'][0], Pm['w_dec'][0])
            part = nll.sum() / N
            part.backward()
            d_out[s:s + chunk] = oc.grad
            loss += float(part.detach())
        out.backward(d_out)
        opt.step()
        return loss
    t0 = time.perf_counter()
    batch = prepare(B, T)
    t_prep = time.perf_counter() - t0
    step(*batch)                                # untimed full-shape warm-up step
    each, loss = [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        loss = step(*batch)
        each.append(time.perf_counter() - t0)
    dt = sorted(each)[len(each) // 2]
    return dict(value=B * T / dt, unit="timesteps/s", cores=cores, kind="port", seconds=dt, seconds_each=each, prepare_seconds=t_prep,
                loss=loss,
                sample=f"median of {reps} train steps (after 1 untimed full-shape step) of C2 [B={B},T={T},88,5] joint LSTM-NADE "
                       f"(oracle/torch_ref.py, float32, {cores} threads, NADE in 2048-row chunks); batch synthesis, dropout uniforms and "
                       f"tensor conversion prepared outside the timed region ({t_prep:.2f} s)")


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
    # the scan is a serial chain per generated step (D / 16 + draws-of-1 passes of one wave per (row, track): nade_sample_chunk_kernel; then two
    # LSTM steps and a Dense step of ~10 us each): its bound is the latency of dependent instructions, not HBM or MFMA; the algorithmic work per
    # generated row is D*Hn MACs + (1 + draws of 1) * Hn sigmoids, reported against the f32 vector peak
    fma = n * steps * (D * Hn + dens * D * Hn)
    res["roofline"] = {"bound": "valu-latency", "achieved_tflops": 2 * fma / t / 1e12, "peak_tflops": PEAK_VALU_F32_TFLOPS,
                       "frac": 2 * fma / t / 1e12 / PEAK_VALU_F32_TFLOPS,
                       "note": "n = 72 rows occupy 72 waves of 1024 SIMD slots: the scan is latency-bound by construction (replicas fill the chip)"}
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


NADE_ENTRIES = ("mnn_nade_logprob", "mnn_density_gate", "mnn_nade_f32_pack")


def step_roofline(N, D, precision, t_step_s, sig_peak, phase_ms, nnz_row, nade_fwd_form):
    """Per phase: measured time (HIP events, eager steps run right behind the timed replays), T_min and frac under two accountings.

    Phase 1, LSTM + Dense (forward and backward, plumbing included in its measured time): 3 x 2 x [(D+R1) 4R1 + (R1+R2) 4R2 + R2 (D+Hn)]
      FLOP per row on MFMA; bytes: uint8 input, saved LSTM activations written + read, weights / gradient / Adam slots.  The kernels execute
      exactly this work (the GEMMs are dense whatever the input density): dense == executed.
    Phase 2, NADE scan (forward and backward):
      dense     SURVEY 8(d): D Hn + D sigmoids forward, D Hn recomputed backward, against the sigmoid rate MEASURED here -- what a kernel that
                evaluated every hidden state of the reference formulation would need.  The kernels do NOT: `a` only moves where v = 1.
      executed  per row with nnz active visibles: sigmoids 2 (1 + nnz) Hn + 2 D; vector FMAs: decoder dots D Hn forward (0 when they run on
                the matrix cores) + 2 D Hn backward + 2 nnz Hn encoder adds; MFMA flops 2 x 2 D Hn when the forward is the matrix-core form (x 3 for the fp16 mode's hi + lo operand pairs).
                T_min = FMAs / f32 vector peak + sigmoids / measured sigmoid rate + MFMA flops / MFMA peak (they share the SIMDs: additive).
    roofline.step.frac = sum of the EXECUTED T_min over the measured step time; "invalid" when any executed phase fraction exceeds 1."""
    R1, R2 = UNITS
    fwd_flop = 2.0 * ((D + R1) * 4 * R1 + (R1 + R2) * 4 * R2 + R2 * (D + HN))
    flops = 3.0 * fwd_flop * N
    peak_mfma = (PEAK_MFMA_F32_TFLOPS if precision == "fp32" else PEAK_MFMA_16_TFLOPS) * 1e12      # f16 and bf16 MFMA: the same dense rate
    act = 4 if precision == "fp32" else 2
    bytes_dense = N * D + 2.0 * N * (4 * R1 + 4 * R2) * act + 2.0 * N * (R1 + R2) * 4 + 2.0 * N * (2 * R1 + 2 * R2) * act + 7 * 4 * N_PARAMS
    bytes_nade = N * (3 * 4 * (HN + D) + 2 * 4 * (HN + D) + D)
    t_mfma, t_hbm1 = flops / peak_mfma, bytes_dense / (PEAK_HBM_GBS * 1e9)
    t1 = max(t_mfma, t_hbm1)
    m1, m2 = phase_ms["lstm_dense"] * 1e-3, phase_ms["nade_scan"] * 1e-3
    # dense accounting of the scan
    sig_dense = N * (2.0 * D * HN + D)
    t2_dense = max(sig_dense / sig_peak, bytes_nade / (PEAK_HBM_GBS * 1e9))
    # executed accounting of the scan
    sig_exec = N * (2.0 * (1.0 + nnz_row) * HN + 2.0 * D)
    on_mfma = nade_fwd_form.startswith("mfma")
    fma_exec = N * ((0.0 if on_mfma else 1.0) * D * HN + 2.0 * D * HN + 2.0 * nnz_row * HN)
    # the fp16 mode's split-operand form issues THREE 16-bit products per logit (hi.hi + hi.lo + lo.hi)
    mfma_exec = N * 4.0 * D * HN * (3.0 if nade_fwd_form == "mfma-split3" else 1.0) if on_mfma else 0.0
    t2_exec = max(fma_exec / (PEAK_VALU_F32_TFLOPS * 1e12 / 2) + sig_exec / sig_peak + mfma_exec / (PEAK_MFMA_16_TFLOPS * 1e12),
                  bytes_nade / (PEAK_HBM_GBS * 1e9))
    phases = {
        "lstm_dense": {"measured_ms": m1 * 1e3, "mfma_ms": t_mfma * 1e3, "hbm_ms": t_hbm1 * 1e3, "flop": flops, "bytes": bytes_dense,
                       "t_min_ms": t1 * 1e3, "frac": t1 / m1 if m1 > 0 else None, "accounting": "dense == executed"},
        "nade_scan": {"measured_ms": m2 * 1e3, "hbm_ms": bytes_nade / (PEAK_HBM_GBS * 1e9) * 1e3, "bytes": bytes_nade, "nnz_per_row": nnz_row,
                      "forward_form": nade_fwd_form,
                      "dense": {"sigmoids": sig_dense, "t_min_ms": t2_dense * 1e3, "frac": t2_dense / m2 if m2 > 0 else None},
                      "executed": {"sigmoids": sig_exec, "valu_fma": fma_exec, "mfma_flop": mfma_exec, "t_min_ms": t2_exec * 1e3,
                                   "frac": t2_exec / m2 if m2 > 0 else None}},
    }
    phases["nade_scan"]["dense"]["exceeds_1"] = bool(m2 > 0 and t2_dense / m2 > 1.0)
    fr = [phases["lstm_dense"]["frac"], phases["nade_scan"]["executed"]["frac"]]
    t_min = t1 + t2_exec
    return dict(t_min_ms=t_min * 1e3, t_measured_ms=t_step_s * 1e3, frac=t_min / t_step_s, invalid=any(f is not None and f > 1.0 for f in fr),
                t_min_dense_ms=(t1 + t2_dense) * 1e3, frac_dense=(t1 + t2_dense) / t_step_s,
                frac_dense_note="SURVEY 8(d)'s dense sigmoid count prices work the kernels do not execute; it exceeds the measured scan time "
                                "when nade_scan.dense.exceeds_1 is true and is reported for reference only",
                phases=phases, accounting="executed (device-counted density); dense beside it")


def _recorded(workload, precision, kind):
    """The newest committed counter recording of this workload / precision (profiles/roundN_<workload>_<precision>_<kind>.json) and whether it
    was taken on THIS build (its sources_sha16 == profiles/tools/source_hash.py of the tree bench.py runs from)."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
    try:
        from source_hash import source_hash
        here = source_hash()
    except Exception:
        here = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"round*_{workload}_{precision}_{kind}.json")),
                   key=lambda f: int(os.path.basename(f)[5:].split("_")[0]))
    for f in reversed(files):
        try:
            data = json.load(open(f))
        except (OSError, ValueError):
            continue
        return data, os.path.basename(f), bool(here and data.get("sources_sha16") == here), here
    return None, None, False, here


def attach_counters(rf, workload, precision):
    """rocprofv3 counters cannot be collected from inside this process: the line carries the counters RECORDED by profiles/tools/record_round6.sh (record_round<N>.sh of the round that made them)
    (separate --pmc passes over this command in eager mode) under roofline.recorded_counters, with the file, the source hash of the recorded
    build and `same_build`.  Only when the recording was taken on this very build are its HBM bytes also reported as roofline.traffic (the
    contract's field); otherwise traffic stays null -- stale counters never stand beside live timings unmarked."""
    kern = rf.get("kernel")
    rf.setdefault("traffic", None)
    if not kern:
        return
    rec = {}
    # gated pairs of launches (both instantiations of a kernel are launched, a device word lets one leave at its first instruction -- the NADE
    # forwards, nade_bwd_kernel<.., USEU>): the one that left moved no data and must not be averaged in as a launch
    tdata = _recorded(workload, precision, "pmc_traffic")[0]
    idle = {k for k, v in (tdata or {}).get("kernels", {}).items() if k.startswith(kern) and v.get("hbm_side_bytes_per_launch", 0.0) < 1e6}
    for kind in ("sq_counters", "pmc_traffic"):
        data, fn, same, here = _recorded(workload, precision, kind)
        if data is None:
            continue
        hit = [v for k, v in data.get("kernels", {}).items() if k.startswith(kern) and k not in idle]
        if not hit:
            continue
        calls = sum(h["calls"] for h in hit)
        r = rec.setdefault(kind, {"file": "profiles/" + fn, "recorded_sources_sha16": data.get("sources_sha16"), "this_build_sha16": here,
                                  "same_build": same})
        if kind == "sq_counters":
            for key in ("mfma_busy_frac", "parked_frac", "active_frac"):
                vals = [(h[key], h["calls"]) for h in hit if key in h]
                if vals:
                    r[key.replace("_frac", "")] = sum(v * c for v, c in vals) / sum(c for _, c in vals)
            r["method"] = "one rocprofv3 --pmc pass of SQ / GRBM counters over this command in eager mode"
        else:
            r["traffic"] = sum(h["hbm_side_bytes_per_launch"] * h["calls"] for h in hit) / calls
            r["traffic_unit"] = "bytes/launch"
            r["traffic_write"] = sum(h["write_bytes"] * h["calls"] for h in hit) / calls
            r["method"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 (MI355X_MICROARCH.md)"
            if same:
                rf["traffic"] = r["traffic"]
                rf["traffic_unit"] = "bytes/launch"
                if rf.get("avg_launch_us"):
                    rf["hbm_gbs"] = r["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9
    if rec:
        rf["recorded_counters"] = rec


def step_traffic(workload, precision, algorithmic_bytes):
    """HBM-side bytes of the whole step (sum over every kernel of the recorded FETCH / WRITE passes) next to the algorithmic bytes of SURVEY 8(d)."""
    out = {"algorithmic_gb_per_step": algorithmic_bytes / 1e9, "hbm_side_gb_per_step": None}
    data, fn, same, here = _recorded(workload, precision, "pmc_traffic")
    if data is not None:
        per_step = data.get("hbm_side_bytes_per_step")
        if per_step is None:         # recordings of earlier rounds: the steps they contain = their clip_adam_kernel launches
            steps = next((k["calls"] for n, k in data["kernels"].items() if n.startswith("clip_adam_kernel")), 4)
            per_step = sum(k["hbm_side_bytes_per_launch"] * k["calls"] for k in data["kernels"].values()) / float(steps)
        out.update(hbm_side_gb_per_step=per_step / 1e9, ratio=per_step / algorithmic_bytes, file="profiles/" + fn, same_build=same,
                   recorded_sources_sha16=data.get("sources_sha16"), this_build_sha16=here)
    return out


def collective_only(a):
    """Launcher / rendezvous rehearsal without a GPU (tests/test_bench_launch.py): N ranks, `gloo`, the flat gradient buffer of the
    joint LSTM-NADE model all-reduced K times with the same barrier + max-over-ranks timing protocol as the real run."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        dist.init_process_group(a.backend if a.backend != "nccl" else "gloo")
    grad = torch.full((N_PARAMS,), float(rank + 1))
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
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.precision], "data": "synthetic",
                          "config": {"workload": w["name"], "parallelism": f"dp{world}"}, "collective_only": True,
                          "allreduce_ok": ok, "allreduce_bytes": grad.numel() * 4, "ranks": world}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def mode_params(w):
    """config / params dictionaries of the reference's YAML files (default_config.yaml, default_params.yaml) for a mode workload."""
    tracks = ["Drums", "Piano", "Guitar", "Bass", "Strings"][:w["M"]]
    config = {"model_name": "bench", "data": {"pitch_range": {"lowest": 0, "highest": w["P"]}, "instruments": tracks, "beat_resolution": 4},
              "training": {"num_pixels": 1, "random_seed": 23}}
    if w["mode"] == "jamming":        # multinn_jamming.py:40-68: per-track RnnRBM on PassEncoder codes, CD-10 (rbm.py default k)
        params = {"mode": "jamming", "tune_encoder": False, "keep_prob": 0.9, "encoder": {"type": "Pass", "num_hidden": None},
                  "generator": {"type": "RBM", "num_hidden": HN, "num_hidden_rnn": list(UNITS), "feedback": None}}
    else:                             # multinn_composer.py:49-87: per-track DBNEncoder [168, 84] -> one LSTM + 5 NADEs over the stacked codes
        params = {"mode": "composer", "tune_encoder": False, "keep_prob": 0.9, "encoder": {"type": "DBN", "num_hidden": [168, 84]},
                  "generator": {"type": "NADE", "num_hidden": HN, "num_hidden_rnn": list(UNITS), "feedback": None}}
    return config, params


def mode_roofline(w, precision, per, calls, N, sec):
    """`roofline` of a mode workload (C3 jamming, C4 composer): the entry point with the largest share of the eager step (HIP events around
    every C-ABI call) against the bound of ITS kernel, from algorithmic counts per launch:
      LSTM recurrences (persist / cluster / resident, single or multi-job)  MFMA: 2 rows (u 4u) per layer and direction;
      mnn_rbm_gibbs*   CD-k chain (rbm.py:148-226): 2 k 2 D Hn flop per row on the f32 matrix cores, k (D + Hn) sigmoids and as many Philox draws;
      mnn_nade_logprob_fwd* / _bwd   HBM: the Dense-output rows the scan reads / writes (as on the joint step);
      mnn_gemm_tn      MFMA: every plain GEMM of the step together (3 x the forward's dense contractions minus the recurrences).
    `others` carries the next entry points the same way."""
    M, P = w["M"], w["P"]
    R1, R2 = UNITS
    jam = w["mode"] == "jamming"
    peak16, peak32 = (PEAK_MFMA_F32_TFLOPS if precision == "fp32" else PEAK_MFMA_16_TFLOPS), PEAK_MFMA_F32_TFLOPS
    stacks = M if jam else 1                                   # jamming: one LSTM per track; composer: one shared LSTM
    E = 84                                                     # composer: DBN code width per track (mode_params)
    d_in = P if jam else E * M
    Dn, tracks = (P, 1) if jam else (E, M)

    def one(name, ms, n):
        r = dict(entry_point=name, launches_per_step=n, avg_launch_us=ms * 1e3 / max(n, 1), measured_ms=ms)
        if name.startswith(("mnn_lstm2_persist", "mnn_lstm_cluster", "mnn_lstm_resident", "mnn_lstm_rowpar")):
            us = {"mnn_lstm_cluster": [R1], "mnn_lstm_resident": [R2]}.get(name.rsplit("_", 2)[0] if name.endswith("_multi") else name.rsplit("_", 1)[0], [R1, R2])
            flop = 2.0 * N * stacks * sum(u * 4 * u for u in us)
            r.update(bound="mfma", achieved=flop / (ms * 1e-3) / 1e12, peak=peak16, unit="TFLOP/s", algorithmic_flop_per_launch=flop / max(n, 1),
                     kernel=name.replace("mnn_", "") + "_kernel", avg_timestep_us=ms * 1e3 / max(n, 1) / w["T"],
                     note="latency-bound chain of T dependent steps (one launch per layer and direction; `_multi`: all tracks' layers in one launch)")
        elif name.startswith("mnn_rbm_gibbs"):
            k = 10
            flop = 2.0 * k * 2 * P * HN * N * M
            r.update(bound="mfma", achieved=flop / (ms * 1e-3) / 1e12, peak=peak32, unit="TFLOP/s", algorithmic_flop_per_launch=flop / max(n, 1),
                     kernel="rbm_gibbs_mfma_kernel", sigmoids_per_launch=k * (P + HN) * N * M / max(n, 1),
                     note="f32 products on v_mfma_f32_32x32x2_f32 (bit-exact draws need the k-ordered f32 chain); Philox + deterministic sigmoid on the VALU "
                          "are ~60 % of an iteration")
        elif name.startswith(("mnn_rbm_hidden", "mnn_rbm_visible", "mnn_rbm_free_energy")):
            flop = 2.0 * 2 * (P * 168 + 168 * E) * N * M if not jam else 2.0 * 2 * P * HN * N * M
            r.update(bound="mfma", achieved=flop / (ms * 1e-3) / 1e12, peak=peak32, unit="TFLOP/s", algorithmic_flop_per_launch=flop / max(n, 1),
                     kernel=name.replace("mnn_", "") + "_kernel")
        elif name.startswith("mnn_nade_logprob"):
            byt = N * (4.0 * tracks * (HN + Dn) * (2 if name.endswith("bwd") else 1) + tracks * Dn + 4.0 * tracks * HN)
            r.update(bound="hbm", achieved=byt / (ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", algorithmic_bytes_per_launch=byt / max(n, 1),
                     kernel="nade_fwd_kernel" if "fwd" in name else "nade_bwd_kernel",
                     note="dense visibles (DBN codes): the f32 vector scan; HBM is not its limiter (one sigmoid per hidden unit and active visible is)")
        else:
            fwd = 2.0 * N * stacks * (d_in * 4 * R1 + R1 * 4 * R2) + 2.0 * N * (M * R2 * (HN + P) if jam else R2 * M * (HN + E))
            flop = 3.0 * fwd
            r.update(bound="mfma", achieved=flop / (ms * 1e-3) / 1e12, peak=peak16, unit="TFLOP/s", algorithmic_flop_per_launch=flop / max(n, 1),
                     kernel="gemm_tn_glds", note="all plain GEMMs of the step (input projections, Dense / Wuh,Wuv layers, their input and weight gradients)")
        r["frac"] = r["achieved"] / r["peak"]
        r["traffic"] = None
        return r

    top = sorted(per.items(), key=lambda kv: -kv[1])
    roof = one(top[0][0], top[0][1], calls[top[0][0]])
    roof["others"] = [one(k, ms, calls[k]) for k, ms in top[1:5] if ms > 0.04 * sum(per.values())]
    roof["sum_device_eager_ms"] = sum(per.values())
    roof["timing"] = "HIP events around every C-ABI call of eager steps run right after the timed replays"
    return roof


def main(argv=None):
    a = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a.gpus, sys.argv[1:] if argv is None else argv, need_devices=not a.collective_only and a.backend == "nccl")
    if a.collective_only:
        return collective_only(a)

    import numpy as np
    import torch
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    ndev = torch.cuda.device_count()
    if world > 1 and a.backend == "nccl" and ndev < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        print(f"# rank {rank}: {ndev} device(s) visible for {world} ranks -- refusing to share a GPU between ranks", file=sys.stderr)
        return 2
    local = local % max(1, ndev)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rehearsal = os.environ.get("MULTINN_DP_REHEARSAL") == "1" and "RANK" in os.environ      # 1-rank RCCL run of the N>1 path
    multi = world > 1 or rehearsal
    dp_info = {"ranks": world, "backend": None, "devices": [local]}
    if multi:
        import torch.distributed as dist
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries ONE JSON line: the C-level descriptor
        # points at stderr until the first collectives have run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(a.backend)
            # what the job really is: the world size after an actual all-reduce over the backend, and every rank's device
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            ids = [torch.zeros(1, device=dev, dtype=torch.int64) for _ in range(dist.get_world_size())]
            dist.all_gather(ids, torch.tensor([local], device=dev, dtype=torch.int64))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        dp_info = {"ranks": dist.get_world_size(), "allreduce_sum_of_ones": float(probe), "backend": dist.get_backend(),
                   "devices": [int(t) for t in ids], "rccl_ranks": dist.get_world_size() if a.backend == "nccl" else None,
                   # which entry point issues the step's ONE all-reduce of the flat gradient (multinn_amd/training.py allreduce_flat)
                   "collective": "mnn_allreduce_flat (C ABI)" if os.environ.get("MULTINN_COMM") == "capi" else "torch.distributed.all_reduce"}
    from multinn_amd import RnnNade, AdamOptimizer, _lib

    w = WORKLOADS[a.workload]
    B, T, P, M = w["B"], w["T"], w["P"], w["M"]
    D = P * M
    N = B * T

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def time_region(step_fn, steps):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step_fn()
        barrier()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        return dt / steps, loss

    # ---- the other BASELINE configurations: the mode classes' captured train step ------------------------------------------------------
    if "mode" in w:
        from multinn_amd import MultINN
        config, params = mode_params(w)
        model = MultINN(config, params, mode=w["mode"], precision=a.precision, seed=23, device=dev)
        model.row0 = rank * B
        x = torch.from_numpy(synth(B, T, P, M, 23 + rank, a.rho)).to(dev)
        opt = AdamOptimizer(0.01)
        for _ in range(max(a.warmup, 1)):
            model.train_step(x, None, opt)
        step_fn, launch = (lambda: model.train_step(x, None, opt)), "eager"
        if not a.no_graph and not multi:
            try:
                step_fn, launch = model.graphed_train_step(x, opt, warmup=1), "hipgraph-replay"
            except Exception as e:
                print(f"# hipGraph capture of the mode's step failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                torch.cuda.synchronize()
        sec, loss = time_region(step_fn, a.steps)
        skipped = sum(int(g.store.skipped) for g in model._generators if getattr(g, "store", None) is not None and g.store.theta is not None)
        model.check(tolerate_overflow=True)          # (f16: a skipped step is the dynamic loss scale at work; it is reported below)
        _lib.TIMING = {}
        nb = min(a.steps, 3)
        for _ in range(nb):
            model.train_step(x, None, opt)
        torch.cuda.synchronize()
        timing, _lib.TIMING = _lib.TIMING, None
        if rank == 0:
            per = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) / nb for k, v in timing.items()}
            calls = {k: len(v) // nb for k, v in timing.items()}
            top = sorted(per.items(), key=lambda kv: -kv[1])
            print(json.dumps({
                "metric": "piano-roll timesteps/sec (train step), " + ("5 x LSTM-RBM CD-10 (jamming)" if w["mode"] == "jamming" else "DBNEncoder -> LSTM-MultiNADE (composer)"),
                "value": world * B * T / sec, "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.precision], "data": "synthetic",
                "config": {"workload": w["name"], "global_batch": world * B, "per_gpu_batch": B, "seq_len": T, "pitches": P, "tracks": M, "rho": a.rho,
                           "parallelism": f"dp{world}"},
                "launch": launch, "loss": float(loss), "optimizer_steps_skipped": skipped, "dp": dp_info, "breakdown_ms": {k: round(v, 3) for k, v in top},
                "roofline": mode_roofline(w, a.precision, per, calls, B * T, sec)}), flush=True)
        if multi:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    def timed_steps(precision, rho, batch, steps, warmup, keep=False, lengths=None):
        """`steps` optimiser steps of a fresh generator on a resident synthetic batch: W eager warm-up steps, capture, K replays
        bracketed by barrier + synchronize, MAX over ranks.  Returns (seconds per step, loss, kept objects or None, launch mode)."""
        gen = RnnNade(D, HN, UNITS, keep_prob=0.9, precision=precision, seed=23, device=dev)
        gen.row0 = rank * batch                               # RNG streams keyed by the GLOBAL sequence index
        x = torch.from_numpy(synth(batch, T, P, M, 23 + rank, rho)).to(dev)
        opt = AdamOptimizer(0.01)
        for _ in range(warmup):
            gen.train_step(x, lengths, opt)
        step_fn = lambda: gen.train_step(x, lengths, opt)
        # hipGraph replay of the step (N>1: forward+backward | eager all-reduce | clip+Adam); a ragged step is captured too in the 16-bit modes:
        # its row counts live on the device (RnnNade.graphed_train_step(lengths=...))
        graph = not a.no_graph and (lengths is None or (precision != "fp32" and gen.ragged_compact))
        if graph:
            try:
                step_fn = gen.graphed_train_step(x, opt, warmup=1, lengths=lengths)
            except Exception as e:      # keep the run alive: fall back to eager launches and say so in the JSON line
                print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                graph = False
                torch.cuda.synchronize()
        sec, loss = time_region(step_fn, steps)
        gen._stack.check()              # a persistent launch that gave up on a bounded spin would have produced garbage: fail loudly
        if keep:
            # SURVEY 8(d) asks for the MEDIAN step: `steps` more replays, each between two HIP events (the contract's value stays the mean of
            # the bracketed region above)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            for e0, e1 in evs:
                e0.record()
                step_fn()
                e1.record()
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            timed_steps.median_ms = ts[len(ts) // 2]
        res = (sec, float(loss), (gen, x, opt) if keep else None, "hipgraph-replay" if graph else "eager")
        if not keep:
            del gen, x, step_fn
            torch.cuda.empty_cache()
        return res

    # ---- the timed region: K steps of the named workload, weak scaling -------------------------------------------------------
    sec, loss, kept, launch = timed_steps(a.precision, a.rho, B, a.steps, a.warmup, keep=True)
    gen, x, opt = kept
    nnz_row = float(x.sum()) / (B * T)            # active visibles per row, counted on the device (a prefix of them are targets: T-1 of T steps)
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
    peak_mfma = PEAK_MFMA_F32_TFLOPS if a.precision == "fp32" else PEAK_MFMA_16_TFLOPS
    rec_flops = 2.0 * N * (R1 * 4 * R1 + R2 * 4 * R2)          # the T sequential [B,u]x[u,4u] products of both layers, one direction
    nade_mfma_form = gen._nade_mfma() and any(k.startswith("mnn_nade_logprob_fwd_mfma") and v[0] > 0.2 for k, v in per_call.items())

    def entry_roofline(dom, dom_ms, dom_calls):
        if dom.startswith("mnn_nade_logprob"):
            bwd = "bwd" in dom
            byts = N * (D + 4 * (HN + D) * (2 if bwd else 1) + 4 * (HN + D if bwd else D))
            kern = "nade_bwd_kernel" if bwd else ("nade_fwd_mfma_kernel" if "mfma" in dom else "nade_fwd_kernel")
            roof = dict(bound="hbm", achieved=byts / (dom_ms * 1e-3) / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", traffic=None,
                        kernel=kern, entry_point=dom, launches_per_step=1, avg_launch_us=dom_ms * 1e3, algorithmic_bytes_per_launch=byts,
                        note="VALU-bound scan (SURVEY 8d): HBM is the contract's bound for a non-MFMA kernel and is NOT its limiter; its "
                             "executed-work fraction is roofline.step.phases.nade_scan.executed")
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
            # (256-unit layers run on the CU-resident kernels instead: their products are counted there)
            res_layers = [u for u in UNITS if u == 256] if any(k.startswith("mnn_lstm_resident") for k in per_call) else []
            direction = "bwd" if dom.endswith("bwd") else "fwd"
            res_layers += [u for u in UNITS if u == 512] if ("mnn_lstm_cluster_" + direction) in per_call else []
            rp_flops = rec_flops - 2.0 * N * sum(u * 4 * u for u in res_layers)
            roof = dict(bound="mfma", achieved=rp_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="lstm_rowpar_%s" % ("bwd" if dom.endswith("bwd") else "fwd"), entry_point=dom, launches_per_step=dom_calls,
                        avg_launch_us=dom_ms * 1e3 / dom_calls, avg_timestep_us=dom_ms * 1e3 / dom_calls / T,
                        algorithmic_flop_per_launch=rp_flops / dom_calls,
                        note="latency-bound chain: per timestep every row tile exchanges its 32-row h (backward: dz) slice with the other unit "
                             "tiles through L2 (flag per wave); the number to watch is avg_timestep_us")
        elif dom in ("mnn_lstm_cluster_fwd", "mnn_lstm_cluster_bwd"):
            # one launch per 512-unit LAYER: eight CUs share 32 rows, 64 units' recurrent weights in each CU's registers, h / dz through the XCD's L2
            cl_flops = 2.0 * N * sum(u * 4 * u for u in UNITS if u == 512)
            roof = dict(bound="mfma", achieved=cl_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="lstm_cl_%s_kernel" % ("bwd" if dom.endswith("bwd") else "fwd"), entry_point=dom, launches_per_step=dom_calls,
                        avg_launch_us=dom_ms * 1e3 / dom_calls, avg_timestep_us=dom_ms * 1e3 / dom_calls / T,
                        algorithmic_flop_per_launch=cl_flops / dom_calls,
                        note="a timestep is 64 MFMAs per wave (every column a distinct row) + the gate pointwise + one exchange of the 32-row tile "
                             "between the cluster's eight CUs (store -> flag -> poll -> LDS-DMA pull); the number to watch is avg_timestep_us")
        elif dom in ("mnn_lstm_resident_fwd", "mnn_lstm_resident_bwd"):
            # one launch per 256-unit LAYER: four batch rows per workgroup, the layer's whole recurrent matrix in the registers + LDS of its CU
            rs_flops = 2.0 * N * sum(u * 4 * u for u in UNITS if u == 256)
            roof = dict(bound="mfma", achieved=rs_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel="lstm_res_%s_kernel" % ("bwd" if dom.endswith("bwd") else "fwd"), entry_point=dom, launches_per_step=dom_calls,
                        avg_launch_us=dom_ms * 1e3 / dom_calls, avg_timestep_us=dom_ms * 1e3 / dom_calls / T,
                        algorithmic_flop_per_launch=rs_flops / dom_calls,
                        note="no hand-offs: a timestep is 128 MFMAs per wave (12 of their 16 columns repeat the workgroup's four rows: the matrix "
                             "streams through the matrix cores once per step whatever the row count) + the gate pointwise; issue-bound, the "
                             "number to watch is avg_timestep_us")
        elif dom in ("mnn_lstm_seq_fwd", "mnn_lstm_seq_bwd", "mnn_lstm2_seq_fwd", "mnn_lstm2_seq_bwd"):
            fused = dom.startswith("mnn_lstm2")
            launches = (T + 2) if fused else 2 * T            # fused: one three-stage launch per timestep for both layers (lag 2)
            roof = dict(bound="mfma", achieved=rec_flops / (dom_ms * 1e-3) / 1e12, peak=peak_mfma, unit="TFLOP/s", traffic=None,
                        kernel=("lstm3_%s_step" if fused else "lstm_%s_step") % ("bwd" if dom.endswith("bwd") else "fwd"),
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
        attach_counters(roof, a.workload, a.precision)
        return roof

    # the contract's `roofline` is for the dominant KERNEL: mnn_gemm_tn launches four different GEMM kernels in a step (gemm_tn_glds256<true|false>,
    # gemm_bres<7|8>, eight launches) whose sum is within box noise of the one nade_bwd_kernel launch -- entries are ranked by their time per
    # distinct kernel so the line does not flip between boxes; the GEMM family follows under `others` with its whole time
    def per_kernel(kv):
        return -kv[1][0] / (4.0 if kv[0] == "mnn_gemm_tn" and kv[1][1] >= 4 else 1.0)
    top = sorted(per_call.items(), key=per_kernel)
    dom, (dom_ms, dom_calls) = top[0]
    roof = entry_roofline(dom, dom_ms, dom_calls)
    roof["others"] = [entry_roofline(k, ms, n) for k, (ms, n) in sorted(top[1:], key=lambda kv: -kv[1][0])[:5] if ms > 0.04 * total_ms]
    roof["timing"] = "HIP events around every C-ABI call of %d eager steps run right after the timed replays" % nb

    sig = sigmoid_peak(dev)
    phase_ms = {"nade_scan": sum(v[0] for k, v in per_call.items() if k.startswith(NADE_ENTRIES)),
                "lstm_dense": sum(v[0] for k, v in per_call.items() if not k.startswith(NADE_ENTRIES))}
    roof["step"] = step_roofline(N, D, a.precision, sec, sig["sigmoids_per_s"], phase_ms, nnz_row,
                                ("mfma-split3" if a.precision == "fp16" else "mfma") if nade_mfma_form else "valu")
    roof["step"]["sigmoid_peak"] = sig
    roof["step"]["rho"] = a.rho
    # algorithmic bytes per step as SURVEY.md 8(d) counts them: x u8 | LSTM saved activations (gates + c + h, 16-bit, written + read) | b_enc, b_dec
    # f32 written + read | conditionals out | weights, gradients, Adam slots  (~6.9 GB at [1024,256,88,5])
    survey_bytes = N * D + 2.0 * 2 * N * 6 * (R1 + R2) + 2.0 * 4 * N * (HN + D) + 4.0 * N * D + 7 * 4 * N_PARAMS
    traffic_step = step_traffic(a.workload, a.precision, survey_bytes)

    out = {
        "metric": "piano-roll timesteps/sec (train step), 5-track LSTM-NADE", "value": world * B * T / sec,
        "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec * 1e3,
        "ms_per_step_median": getattr(timed_steps, "median_ms", None),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[a.precision],
        "data": "synthetic",
        "config": {"workload": w["name"], "global_batch": world * B, "per_gpu_batch": B, "seq_len": T, "pitches": P, "tracks": M, "rho": a.rho,
                   "nade_hidden": HN, "lstm_units": UNITS, "keep_prob": 0.9, "optimizer": "TF-Adam lr 0.01 eps 1e-4 clip 5.0",
                   "parallelism": f"dp{world}"},
        "launch": launch, "loss": loss, "optimizer_steps_skipped": int(gen.store.skipped), "loss_scale_multiplier": float(gen.store.ls_dyn[0]), "dp": dp_info,
        "parity": {"mode": a.precision,
                   "gate": ("tests/test_gpu_realdims.py: loss, per-row NLL <= 1e-4 relative, conditionals <= 1e-4 absolute vs the float64 oracle at D=440, "
                            "Hn=256, [512,256], rho in {0.03, 0.5}, persistent kernels asserted ON") if a.precision in ("fp16", "fp32") else
                           ("bf16 operands (8 significant bits): loss 6e-6, per-row NLL 1.2e-4, conditionals 3e-4 abs, gradients 2e-3..4.5e-3 vs the "
                            "float64 oracle (tests/test_gpu_realdims.py, printed bounds)")},
        "roofline": roof,
        "hbm_traffic": traffic_step,
        "breakdown_ms": {k: round(v[0], 3) for k, v in top},
    }
    out.update(extras)
    if not a.no_sampling:
        out["sampling"] = sampling_scan(gen, P, M)
    del gen, x, opt, kept
    torch.cuda.empty_cache()
    if world == 1 and not rehearsal and not a.no_extras:
        # strong-scaling proxy: the per-GPU share of the GLOBAL batch at 8 GPUs, B / 8 sequences, on this one GPU -- what the step
        # costs on the kernel forms that shape selects; 8 x that throughput is the ceiling of the strong-scaling leg before any collective
        if B % 8 == 0 and (B // 8) % 32 == 0:
            sp, lp, _, lmp = timed_steps(a.precision, a.rho, B // 8, max(3, a.steps // 2), 2)
            out["strong_proxy"] = {"per_gpu_batch": B // 8, "global_batch": B, "ms_per_step": sp * 1e3, "value_1gpu": (B // 8) * T / sp,
                                   "value_8gpu_ceiling": 8 * (B // 8) * T / sp, "unit": "timesteps/s", "vs_weak_per_row": (sp / (B // 8)) / (sec / B),
                                   "launch": lmp, "note": "no collective in it: 8 x the one-GPU rate of the B/8 shape"}
        # dense stress input: the NADE kernels' exact sparsity shortcuts vanish at rho = 0.5
        s5, l5, _, _ = timed_steps(a.precision, 0.5, B, max(3, a.steps // 2), 2)
        out["rho05"] = {"rho": 0.5, "ms_per_step": s5 * 1e3, "value": B * T / s5, "unit": "timesteps/s", "loss": l5,
                        "vs_rho_headline": s5 / sec}
        # ragged batch: lengths ~ U{T/2..T}, seed 24; value counts VALID timesteps only
        ln = ragged_lengths(B, T)
        sr, lr_, _, lm = timed_steps(a.precision, a.rho, B, max(3, a.steps // 2), 2, lengths=torch.from_numpy(ln).to(dev))
        out["ragged"] = {"lengths": "U{T/2..T}, numpy PCG64 seed 24", "valid_timesteps": int(ln.sum()), "ms_per_step": sr * 1e3,
                         "value": float(ln.sum()) / sr, "unit": "valid timesteps/s", "loss": lr_, "launch": lm, "vs_headline_ms": sr / sec,
                         "rows": "Dense + NADE on the valid rows only (ops.ragged_index: compaction, row count and loss scale on the device); "
                                 "the LSTM steps every row (impute_finished=False)"}
        for other in [p for p in ("bf16", "fp32") if p != a.precision]:
            so, lo, _, lm = timed_steps(other, a.rho, B, 3 if other == "fp32" else max(3, a.steps // 2), 1)
            out[other] = {"ms_per_step": so * 1e3, "value": B * T / so, "unit": "timesteps/s", "loss": lo, "launch": lm,
                          "note": {"bf16": "8-bit operands, the matrix-core NADE forward: misses the 1e-4 gate on per-row NLL and conditionals",
                                   "fp32": "v_mfma_f32_32x32x2_f32 GEMMs, launch-per-timestep recurrence, f32 NADE kernels"}[other]}
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(P, M, a.rho)
    print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
