"""Build-time guards on the recurrence kernels: no private (scratch) memory.

The row-parallel LSTM kernels keep a whole timestep's operands in registers and request the next step's operands a step ahead; an array that the
compiler leaves in scratch memory turns those prefetches into store-and-reload round trips (measured: backward 3.8 -> 6.6 ms per train step at the
bench shape when a `float4[4]` staging array went to scratch).  hipcc cross-compiles without a GPU, so this runs in the CPU suite."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_rowpar_kernels_use_no_scratch_memory(tmp_path):
    from multinn_amd import build
    src = os.path.join(build.CSRC, "lstm_rowpar.hip")
    out = str(tmp_path / "rp.s")
    subprocess.check_call([HIPCC] + build.FLAGS + ["-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    text = open(out).read()
    sizes = {m.group(1): int(m.group(2))
             for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    kernels = {k: v for k, v in sizes.items() if "lstm_rowpar_" in k and "kernel" in k}
    assert len(kernels) >= 36, sizes        # three widths x forward (f32 | 16-bit xproj) / backward x (one wave | a wave pair) x (bf16 | f16)
    assert all(v == 0 for v in kernels.values()), kernels
    # The flag of a hand-off waits with `s_waitcnt vmcnt(N)`, N = the vector-memory operations WRITTEN BEHIND the hand-off stores in the
    # source.  That only covers the tile if the compiler keeps them behind: RP_HANDOFF_FENCE (sched_barrier + memory clobber) sits right
    # behind the hand-off stores and leaves a marker.  In every kernel, between the marker and the closest vector-memory instruction in
    # front of it there must be nothing but the hand-off's own 16-byte buffer stores.
    bodies = re.split(r"\n(?=_Z\w*lstm_rowpar_\w+:)", text)
    checked = 0
    for body in bodies:
        if not re.match(r"_Z\w*lstm_rowpar_(fwd|bwd)", body):
            continue
        lines = body.split("\n")
        marks = [i for i, ln in enumerate(lines) if "RP_HANDOFF_FENCE" in ln]
        assert len(marks) >= 1, body[:80]
        for i in marks:
            j = i - 1
            while j >= 0 and not re.search(r"\b(global_load|global_store|global_atomic|buffer_load|buffer_store|flat_load|flat_store|scratch_)", lines[j]):
                j -= 1
            assert j >= 0 and "buffer_store_dwordx4" in lines[j], (body[:60], lines[j] if j >= 0 else None)
            checked += 1
    assert checked >= 36, checked


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_matrix_core_chain_kernels_use_no_scratch_and_read_operands_ahead(tmp_path):
    """The f32 matrix-core chain kernels of round 4 (deterministic LSTM / Dense steps, Gibbs chain, sampling half-steps): no scratch memory
    (their weight rings and operand batches are register arrays with compile-time indices only), and the MFMAs of the det steps (one wave
    per SIMD: nothing else hides an LDS wait) come in runs -- a rolled loop with one LDS read + wait in front of every MFMA would show up as
    single MFMAs separated by `s_waitcnt lgkmcnt(0)`.  (The Gibbs / half-step kernels run two waves per SIMD and keep the rolled loop: measured.)"""
    from multinn_amd import build
    for fn, names, min_run in (("det_step.hip", ("lstm_step_det_kernel", "dense_det_kernel"), 8), ("rbm.hip", ("rbm_gibbs_mfma_kernel", "rbm_half_mfma_kernel"), 1)):
        out = str(tmp_path / (fn + ".s"))
        subprocess.check_call([HIPCC] + build.flags_for(fn) + ["-S", "--cuda-device-only", os.path.join(build.CSRC, fn), "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
        sizes = {m.group(1): int(m.group(2))
                 for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
        for nm in names:
            hit = {k: v for k, v in sizes.items() if nm in k}
            assert hit and all(v == 0 for v in hit.values()), (nm, hit)
        for body in re.split(r"\n(?=_Z\w+:)", text):
            if not any(re.match(r"_Z\w*" + nm, body) for nm in names):
                continue
            runs, cur = [], 0
            for ln in body.split("\n"):
                if "v_mfma_f32_32x32x2_f32" in ln:
                    cur += 1
                elif "s_waitcnt" in ln and "lgkmcnt(0)" in ln and cur:
                    runs.append(cur)
                    cur = 0
            runs.append(cur)
            assert max(runs) >= min_run, (body[:70], runs[:20])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_resident_recurrence_keeps_its_weights_in_registers(tmp_path):
    """The CU-resident recurrence (lstm_resident.hip) holds 384 of a lane's 512 registers of recurrent weights for the whole sequence: every
    instantiation must compile without scratch memory (a spilled fragment is reloaded with `s_waitcnt vmcnt(0)` behind the step's LDS-DMA:
    the whole memory latency on the chain of every timestep), the matrix cores must read the AGPR-pinned fragments in place (no
    v_accvgpr_read in front of an MFMA: the MFMA results live in VGPRs, -amdgpu-mfma-vgpr-form), and the step must stay one scheduling region
    (all 128 MFMAs of a wave's timestep in one basic block of the loop)."""
    from multinn_amd import build
    out = str(tmp_path / "res.s")
    subprocess.check_call([HIPCC] + build.flags_for("lstm_resident.hip") + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "lstm_resident.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    sizes = {m.group(1): int(m.group(2))
             for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    kernels = {k: v for k, v in sizes.items() if "lstm_res_" in k}
    assert len(kernels) == 12, sizes            # forward: 2 flavours x (mask | none) x (saving | not); backward: 2 flavours x (mask | none)
    assert all(v == 0 for v in kernels.values()), kernels
    for body in re.split(r"\n(?=_Z\w+:)", text):
        if not re.match(r"_Z\w*lstm_res_(fwd|bwd)_kernel", body):
            continue
        blocks = re.split(r"\n\.LBB\w+:", body)
        per_block = [len(re.findall(r"v_mfma_f32_16x16x32", b)) for b in blocks]
        assert max(per_block) == 128, (body[:60], per_block)
        assert len(re.findall(r"v_mfma_f32_16x16x32_\w+ \S+ a\[", body)) >= 64, body[:60]      # srcA straight from AGPRs
        assert "v_accvgpr_read" not in "".join(b for b, n in zip(blocks, per_block) if n == 128), body[:60]
        if re.match(r"_Z\w*lstm_res_fwd_kernel", body):
            # The step waits for the NEXT step's LDS-DMA with `s_waitcnt vmcnt(N)`, N = the buffer stores WRITTEN behind the DMA in the source
            # (res_wait_all_but(EMIT_STORES + NG * (SAVE ? 2 : 1))).  The wait covers the DMA only if exactly N vector-memory operations, all of
            # them stores, stand between the last global_load_lds and that wait in the ISA: a merged, elided or extra memory operation would make
            # step t + 1 read rows that have not landed.
            big = [b for b, n in zip(blocks, per_block) if n == 128][0].split("\n")
            dma = [i for i, ln in enumerate(big) if "global_load_lds" in ln]
            assert dma, body[:60]
            behind, wait = [], None
            for ln in big[dma[-1] + 1:]:
                if re.search(r"\b(global_load|global_store|global_atomic|buffer_load|buffer_store|flat_|scratch_)", ln):
                    behind.append(ln.split()[0])
                mm = re.search(r"s_waitcnt vmcnt\((\d+)\)", ln)
                if mm:
                    wait = int(mm.group(1))
                    break
            assert wait is not None and wait == len(behind) and all(x.startswith("buffer_store") for x in behind), (body[:60], wait, behind)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_weight_resident_gemm_uses_no_scratch_and_reads_its_weights_in_place(tmp_path):
    """gemm_bres.hip counts its vector-memory operations by hand (the wait for an LDS-DMA stage allows exactly the younger DMA pieces and C
    stores): a register spilled to scratch memory would add loads / stores of its own to that count and let a stage be read before it has
    landed.  Every instantiation must compile without scratch, and the matrix cores must read the pinned weight fragments from AGPRs."""
    from multinn_amd import build
    out = str(tmp_path / "bres.s")
    subprocess.check_call([HIPCC] + build.flags_for("gemm_bres.hip") + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "gemm_bres.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    sizes = {m.group(1): int(m.group(2))
             for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    kernels = {k: v for k, v in sizes.items() if "gemm_bres_kernel" in k}
    assert len(kernels) >= 4 and all(v == 0 for v in kernels.values()), kernels
    for body in re.split(r"\n(?=_Z\w+:)", text):
        if not re.match(r"_Z\w*gemm_bres_kernel", body):
            continue
        assert "scratch_" not in body, body[:60]
        n_mfma = len(re.findall(r"v_mfma_f32_32x32x16", body))
        assert len(re.findall(r"v_mfma_f32_32x32x16_\w+ \S+ a\[", body)) >= 0.75 * n_mfma, body[:60]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_cluster_recurrence_keeps_its_weights_in_agprs_and_counts_its_staging(tmp_path):
    """lstm_cluster.hip: a wave's 64 weight fragments (256 registers) are AGPR-pinned and read in place by every MFMA (64 per timestep and
    wave, both directions); no scratch memory (the forward waits for its LDS-DMA pull with a counted `s_waitcnt vmcnt(N)`, N = the staging
    requests issued behind it: a spill would add vector-memory operations of its own to that count)."""
    from multinn_amd import build
    out = str(tmp_path / "cl.s")
    subprocess.check_call([HIPCC] + build.flags_for("lstm_cluster.hip") + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "lstm_cluster.hip"), "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    sizes = {m.group(1): int(m.group(2))
             for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text)}
    kernels = {k: v for k, v in sizes.items() if "lstm_cl_" in k}
    assert len(kernels) == 12 and all(v == 0 for v in kernels.values()), kernels         # forward (drop x save) + backward (drop), two flavours
    seen = 0
    for body in re.split(r"\n(?=_Z\w+:)", text):
        m = re.match(r"_Z\w*lstm_cl_(fwd|bwd)_kernel\w*:", body)
        if not m:
            continue
        seen += 1
        assert "scratch_" not in body, body[:60]
        mf = re.findall(r"v_mfma_f32_32x32x16_\w+ \S+ (\S)\[", body)
        assert len(mf) == 64 and all(x == "a" for x in mf), (body[:60], len(mf))
        if m.group(1) == "fwd":
            # the wait in front of the step's barrier: exactly the staging requests (4 xproj row pairs + 2 keep-byte pieces with dropout) may be in flight
            lines = [ln.strip() for ln in body.split("\n")]
            drop = "Lb1ELb" in body[:80]
            waits = [i for i, ln in enumerate(lines) if re.match(r"s_waitcnt vmcnt\((4|6)\)", ln)]
            assert waits, body[:60]
            i = waits[-1]
            assert lines[i] == "s_waitcnt vmcnt(%d)" % (6 if drop else 4), (body[:60], lines[i])
            behind = []
            j = i - 1
            while j >= 0 and len(behind) < (6 if drop else 4):
                if re.match(r"(global_|buffer_|flat_)", lines[j]):
                    behind.append(lines[j].split()[0])
                j -= 1
            assert all(x.startswith("global_load_lds_dword") for x in behind), (body[:60], behind)
    assert seen == 12, seen
