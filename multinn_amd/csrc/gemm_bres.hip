// Weight-resident GEMM for the step's input projections:  C[M, N] = A[M, K] . B[N, K]^T + bias,  M = B*T rows (262 144), K = 448 / 512,
// a 16-bit C  (xproj of LSTM layer 1 and 2: rnn.py:124 through rnn_nade.py:204-218).
//
// Why another form (profiles/round5_a_gemm_pair_notes.md): an LDS-staged tile of these shapes is bounded by three streams it cannot overlap --
// operands through the CU's ~28 B/clk L2 -> LDS path (1 B per 128 FLOP for a 256 x 256 tile), MFMA, and C out.  Here the WEIGHT panel never
// moves: a workgroup (4 waves, one per SIMD, one workgroup per CU) owns 256 columns for its whole life, each wave keeps its 64 columns x K of
// B as MFMA operand fragments in REGISTERS (K / 2 = 224 .. 256 per lane, pinned to the AGPR half; the matrix cores read them in place: the
// idiom of lstm_resident.hip), and only the activation rows stream -- 128 rows x 64 k = 16 KiB per stage through an 8-slot LDS ring, six
// stages in flight, across tile boundaries (the workgroup walks 32 .. 64 row tiles: one prologue per CU, not per tile).  1 B of operand per
// 256 FLOP, no B fragment reads, 4 LDS reads per 8 MFMAs.
// A tile's C leaves from the registers of the NEXT tile's first stage: at the tile's end the accumulators are packed (bias added) into 64
// registers, and their 16 stores are issued behind the next tile's first barrier, so the store path drains under the next tile's MFMAs.
// The MFMA operands are swapped (weights as the A operand) as in gemm_tn_pair_kernel: a lane holds 4 consecutive columns of one row of C.
// vmcnt bookkeeping: LDS-DMA and stores retire in issue order, so the wait for stage n counts the younger operations exactly: 5 stages x 4
// DMA pieces, plus the 16 C stores where the previous tile's epilogue falls inside the window (stages 1 .. 6 of every tile but the first).
#include "common.h"
#include <algorithm>
#include <stdlib.h>
#include <utility>

typedef __attribute__((address_space(1))) const void* br_gas_t;
typedef __attribute__((address_space(3))) void* br_lds_t;
typedef float br_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int br_u4 __attribute__((ext_vector_type(4)));

#define BR_NSTG 8
#define BR_PF 6
#define BR_STG (128 * 128)
#define BR_SC (32 * 144)                              // per-wave scratch of the epilogue: 32 rows x (128 + 16 pad) bytes
#define BR_LDS (BR_NSTG * BR_STG + 256 * 4 + 4 * BR_SC)

template <int N>
__device__ __forceinline__ void br_wait() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// The previous tile's 16 C stores are spread over the stages of the current tile (a burst behind one barrier stalls all four waves at the
// CU's store-issue rate: 190 of 595 us on xproj1): stage q issues stores [br_cum(q), br_cum(q + 1)).
template <int KS> __host__ __device__ constexpr int br_st(int q) { return 16 / KS + (q < 16 % KS ? 1 : 0); }
template <int KS> __host__ __device__ constexpr int br_cum(int q) { int c = 0; for (int i = 0; i < q; ++i) c += br_st<KS>(i); return c; }
// Stores younger than the LDS-DMA of stage `qm` (of the current tile; qm = 0 .. KS - 1) when the workgroup waits for it: those of the 6 stages
// in front of it -- every store of a stage is issued in front of the synchronisation point inside that stage, and that DMA was issued at the
// synchronisation 6 stages back.  cls = min(tile index, 2): the current tile has pending stores from tile 1 on, the previous one from tile 2 on.
template <int KS> __host__ __device__ constexpr int br_window(int qm, int cls) {
    int c = 0;
    for (int d = 1; d <= BR_PF; ++d) {
        const int s = qm - d;
        if (s >= 0) c += cls >= 1 ? br_st<KS>(s) : 0;
        else c += cls >= 2 ? br_st<KS>(s + KS) : 0;
    }
    return c;
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) (stage / group / store indices must be constant expressions:
// they select wait counts and registers)
template <typename Fn, int... I>
__device__ __forceinline__ void br_unroll(Fn&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename Fn>
__device__ __forceinline__ void br_for(Fn&& f) { br_unroll(f, std::make_integer_sequence<int, N>{}); }

// VAR (development, MNN_GEMM_BRES_VAR): 0 product; 1 no C stores; 2 no LDS-DMA in the loop; 3 neither (MFMA + LDS reads + epilogue arithmetic)
template <typename F, int KS, int VAR = 0>
__global__ void __launch_bounds__(256, 1)
gemm_bres_kernel(const h16_t* __restrict__ A, int lda, const h16_t* __restrict__ B, int ldb, h16_t* __restrict__ C, int ldc,
                 const float* __restrict__ bias, int M, int N, int ncp, int nslab, int tiles_per_slab) {
    using frag_t = typename F::x8;
    static_assert(KS >= 7 && KS <= 8, "the epilogue window of the vmcnt bookkeeping assumes 7 or 8 stages per tile");
    constexpr int NK16 = 4 * KS;
    extern __shared__ __attribute__((aligned(16))) char br_smem[];        // [8 stages][128 rows x 128 B] | bias f32 [256]
    float* bias_l = reinterpret_cast<float*>(br_smem + BR_NSTG * BR_STG);
    // the column panels of one row slab run side by side on one XCD (blocks b and b + 8 share one): the slab's rows come out of that L2
    const int bid = blockIdx.x, qq = bid >> 3;
    const int cp = qq % ncp, slab = (qq / ncp) * 8 + (bid & 7);
    if (slab >= nslab) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int colw = cp * 256 + wave * 64;
    const long row_base = (long)slab * tiles_per_slab * 128;
    const int ntile = (int)std::min<long>(tiles_per_slab, ((long)M - row_base) / 128);
    if (ntile <= 0) return;
    const int nstage = ntile * KS;

    // ---- this wave's 64 columns of B as MFMA operand fragments, pinned to AGPRs (rows past N replicate the last: never stored)
    frag_t wr[2][NK16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const h16_t* __restrict__ pb = B + (size_t)(colw + 32 * j + r) * ldb + 8 * h;
#pragma unroll
        for (int kk = 0; kk < NK16; ++kk) wr[j][kk] = *reinterpret_cast<const frag_t*>(pb + 16 * kk);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < NK16; ++kk) asm volatile("" : "+a"(wr[j][kk]));
    bias_l[threadIdx.x] = (bias != nullptr && cp * 256 + (int)threadIdx.x < N) ? bias[cp * 256 + threadIdx.x] : 0.f;

    // ---- the activation stream: this lane's 4 LDS-DMA pieces of a stage (tile-relative element offsets; XOR-swizzled 128-byte rows)
    unsigned off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int p = (s * 4 + wave) * 64 + lane, row = p >> 3, pc = p & 7;
        off[s] = (unsigned)row * (unsigned)lda + (unsigned)((pc ^ ((row >> 1) & 7)) * 8);
    }
    const h16_t* __restrict__ Aslab = A + (size_t)row_base * lda;
    auto issue = [&](int n) {                                 // stage n = (tile n / KS, k stage n % KS); past the end: the last stage again, into a slot nobody reads
        const int nn = std::min(n, nstage - 1), t = nn / KS, q = nn - t * KS;
        const h16_t* b = Aslab + (size_t)t * 128 * lda + q * 64;
        char* dst = br_smem + (n & (BR_NSTG - 1)) * BR_STG;
#pragma unroll
        for (int s = 0; s < 4; ++s)
            __builtin_amdgcn_global_load_lds((br_gas_t)(b + off[s]), (br_lds_t)(dst + (s * 4 + wave) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int n = 0; n < BR_PF; ++n) issue(n);

    const int sw = (r >> 1) & 7;
    const int offr = r * 128;
    uint32_t kxv[4];                                           // byte offset of this lane's 16-byte chunk of k-step ks inside its (swizzled) row
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kxv[ks] = (uint32_t)(((2 * ks + h) ^ sw) << 4);
    br_f32x16 acc[4][2];
    // The previous tile's C, packed and LINE-contiguous: pend[i][p] = 8 consecutive columns (16 bytes) of row 32 i + 8 p + (lane >> 3), piece
    // lane & 7 of the row's 128 bytes -- a store instruction then covers 8 rows x 128 bytes (whole lines).  With a lane's own accumulator
    // quads (32 rows x 32-byte pieces per instruction) the stores cost the CU's memory pipeline four times the address work of an LDS-DMA
    // piece and the activation stream queued behind them: 596 us with the stream + stores against 401 / 429 with either alone.
    br_u4 pend[4][4];
    char* scw = br_smem + BR_NSTG * BR_STG + 256 * 4 + wave * BR_SC;
    const int prow = lane >> 3, pcol = colw + 8 * (lane & 7);
    const uint32_t lane_off = ((uint32_t)prow * (uint32_t)ldc + (uint32_t)pcol) * 2u;
    auto emit1 = [&](int t_prev, auto sc) {                    // store sc (= (i, p)) of tile t_prev
        constexpr int sidx = decltype(sc)::value, i = sidx >> 2, pp = sidx & 3;
        // a wave-uniform 64-bit base (scalar arithmetic) + ONE 32-bit lane offset: sixteen per-lane 64-bit row pointers would be hoisted out of
        // the tile loop (32 registers, spilled to scratch at K = 512 -- and scratch traffic would break the vmcnt bookkeeping)
        char* cb = reinterpret_cast<char*>(C) + (size_t)(row_base + (long)t_prev * 128 + i * 32 + pp * 8) * (size_t)ldc * 2;
        if (VAR & 1) asm volatile("" :: "v"(pend[i][pp]));
        // unconditional (N is a multiple of 256): the vmcnt bookkeeping counts on every store issuing.  NON-TEMPORAL: C is written once and read by a
        // later kernel; stored with the default policy its gigabyte pushes the activation slabs -- which the eight CUs of a column-panel group
        // share through their XCD's L2 -- out before all of them have read them (xproj1 536 -> 504 us, xproj2 313 -> 289 us)
        else __builtin_nontemporal_store(pend[i][pp], reinterpret_cast<br_u4*>(cb + lane_off));
    };

    // sync(qm): stage (t, qm) has landed for every wave (this wave's pieces by a counted vmcnt -- everything younger may still fly: 5 stages
    // of 4 pieces and the stores of the 6 stages before it --, the others' behind the barrier), and the slot of the stage two back is free:
    // the LDS-DMA of the stage 6 ahead goes into it.
    auto sync = [&](int t, auto qmc) {
        constexpr int qm = decltype(qmc)::value;
        constexpr int DMA_YOUNGER = (BR_PF - 1) * 4;
        if (VAR & 2) br_wait<0>();
        else if ((VAR & 1) || t == 0) br_wait<DMA_YOUNGER>();
        else if (t == 1) br_wait<DMA_YOUNGER + br_window<KS>(qm, 1)>();
        else br_wait<DMA_YOUNGER + br_window<KS>(qm, 2)>();
        __builtin_amdgcn_s_barrier();
        if (!(VAR & 2)) issue(t * KS + qm + BR_PF);
    };
    // ring of fragment pairs (one group = 2 row tiles of one k-step): the reads run RD - 1 groups ahead.  K = 512 fills the AGPR half with
    // weights and leaves 8 registers less: one group ahead there
    constexpr int RD = KS == 8 ? 2 : 3, AH = RD - 1;
    frag_t a[RD][2];
    auto rd = [&](uint32_t la, int g, frag_t (&dst)[2]) {
        const uint32_t ad = la + kxv[g >> 1];
        if (g & 1) {
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(dst[0]) : "v"(ad));
            asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(dst[1]) : "v"(ad));
        } else {
            asm volatile("ds_read_b128 %0, %1" : "=v"(dst[0]) : "v"(ad));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(dst[1]) : "v"(ad));
        }
    };
    for (int t = 0; t < ntile; ++t) {
        // A tile = KS stages x 8 groups (k-step, row half) of 2 fragment reads + 4 MFMAs, ONE continuous stream: the reads run two groups
        // ahead of their MFMAs in inline assembly with counted lgkmcnt (hipcc re-uses one register pair and waits lgkmcnt(0) per group: the
        // whole LDS latency per 4 MFMAs, with nothing else on the SIMD to cover it), also ACROSS stage boundaries -- stage q + 1 is
        // synchronised in the middle of stage q (behind group 5), so its first fragments are requested under stage q's last MFMAs.
        // sched_barrier: hipcc moves MFMAs across an asm wait otherwise (cdna_hip_programming.md rule 18).
        sync(t, std::integral_constant<int, 0>{});
        uint32_t la = (uint32_t)(uintptr_t)(br_smem + ((t * KS) & (BR_NSTG - 1)) * BR_STG + offr), la_next = la;
        rd(la, 0, a[0]);
        if constexpr (AH == 2) rd(la, 1, a[1]);
        br_for<KS>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            br_for<8>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int G = 8 * q + g;                   // group index in the tile: ring slot G % 3
                if constexpr (g == 8 - AH && q + 1 < KS) {
                    sync(t, std::integral_constant<int, q + 1>{});
                    la_next = (uint32_t)(uintptr_t)(br_smem + ((t * KS + q + 1) & (BR_NSTG - 1)) * BR_STG + offr);
                }
                if constexpr (g + AH < 8) rd(la, g + AH, a[(G + AH) % RD]);
                else if constexpr (q + 1 < KS) rd(la_next, g + AH - 8, a[(G + AH) % RD]);
                // reads younger than this group's: AH groups ahead (2 each), fewer at the tile's end
                constexpr int ahead = (q + 1 < KS) ? AH : (7 - g < AH ? 7 - g : AH);
                if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                constexpr int ks = g >> 1, i0 = 2 * (g & 1);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (q == 0 && ks == 0) {
                            br_f32x16 z;
#pragma unroll
                            for (int e = 0; e < 16; ++e) z[e] = 0.f;
                            acc[i0 + ii][j] = F::mfma32(wr[j][4 * q + ks], a[G % RD][ii], z);
                        } else {
                            acc[i0 + ii][j] = F::mfma32(wr[j][4 * q + ks], a[G % RD][ii], acc[i0 + ii][j]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
                // the previous tile's stores of this stage, one behind each of the first odd groups (all in front of the stage's synchronisation point)
                if constexpr ((g & 1) && g < 6 && (g >> 1) < br_st<KS>(q)) {
                    if (t >= 1) emit1(t - 1, std::integral_constant<int, br_cum<KS>(q) + (g >> 1)>{});
                }
                if constexpr (g == 7) la = la_next;
            });
        });
        // ---- the tile's C: + bias, rounded, 32 rows at a time through the wave's LDS scratch (written in the accumulator layout: a quad = 8
        // bytes of a row; read back row-contiguous: 16 bytes per lane) into the pending registers.  (LDS operations of one wave execute in order; the
        // compiler keeps the writes in front of the reads of the same buffer.  No asm memory clobber here: it would force `pend` into scratch memory.)
        // the 8 bias quads of the lane's columns, requested together IN FRONT of the scratch writes (the compiler cannot move an LDS load over an
        // LDS store that may alias: behind them each quad costs a round trip).  K = 448 has the registers to read them once per tile; K = 512
        // re-reads them per row tile (32 registers less while acc and pend are both live)
        float4 bq[2][4];
        if constexpr (KS < 8) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) bq[j][qd] = *reinterpret_cast<const float4*>(bias_l + wave * 64 + 32 * j + 8 * qd + 4 * h);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (KS >= 8) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) bq[j][qd] = *reinterpret_cast<const float4*>(bias_l + wave * 64 + 32 * j + 8 * qd + 4 * h);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    uint2 pk;
                    pk.x = pack2<F>(acc[i][j][4 * qd + 0] + bq[j][qd].x, acc[i][j][4 * qd + 1] + bq[j][qd].y);
                    pk.y = pack2<F>(acc[i][j][4 * qd + 2] + bq[j][qd].z, acc[i][j][4 * qd + 3] + bq[j][qd].w);
                    *reinterpret_cast<uint2*>(scw + r * 144 + (32 * j + 8 * qd + 4 * h) * 2) = pk;
                }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) pend[i][pp] = *reinterpret_cast<const br_u4*>(scw + (8 * pp + prow) * 144 + (lane & 7) * 16);
        }
    }
    br_for<16>([&](auto sc) { emit1(ntile - 1, sc); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped re-loads of the last stages are LDS writes: none may outlive the workgroup
}

// K = 448 or 512, 16-bit operands and C, M a multiple of 128, N a multiple of 256 (whole column panels: every store of the epilogue
// issues, which its vmcnt bookkeeping relies on), plain store: the shapes this form is built and measured for
int mnn_gemm_bres_ok(int M, int N, int K) { return ((K == 448 || K == 512) && M >= 128 * 64 && M % 128 == 0 && N >= 256 && N % 256 == 0) ? 1 : 0; }

int mnn_gemm_bres_launch(hipStream_t st, int f16, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const float* bias) {
    MNN_REQUIRE(mnn_gemm_bres_ok(M, N, K), "gemm_bres: shape not covered (M=%d N=%d K=%d)", M, N, K);
    MNN_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 &&
                    ((uintptr_t)C & 15) == 0 && (size_t)128 * 64 * (size_t)lda < ((size_t)1 << 31),
                "gemm_bres: operands 16-byte aligned with pitches that are multiples of 8");
    typedef void (*fn_t)(const h16_t*, int, const h16_t*, int, h16_t*, int, const float*, int, int, int, int, int);
    fn_t fn;
    if (K == 448) fn = f16 ? gemm_bres_kernel<Fp16F, 7> : gemm_bres_kernel<Bf16F, 7>;
    else fn = f16 ? gemm_bres_kernel<Fp16F, 8> : gemm_bres_kernel<Bf16F, 8>;
#ifdef BR_ABLATION      // development builds only (profiles/tools/gemm_bres_probe.py): ablation variants that do NOT store C, chosen by MNN_GEMM_BRES_VAR
    {
        const char* ve = getenv("MNN_GEMM_BRES_VAR");
        const int var = ve ? atoi(ve) : 0;
        if (var >= 1 && var <= 3 && f16) {
            if (K == 448) fn = var == 1 ? gemm_bres_kernel<Fp16F, 7, 1> : (var == 2 ? gemm_bres_kernel<Fp16F, 7, 2> : gemm_bres_kernel<Fp16F, 7, 3>);
            else fn = var == 1 ? gemm_bres_kernel<Fp16F, 8, 1> : (var == 2 ? gemm_bres_kernel<Fp16F, 8, 2> : gemm_bres_kernel<Fp16F, 8, 3>);
            MNN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS));
            const int ncp_ = cdiv(N, 256), mt_ = M / 128;
            int ns_ = std::min(std::max(8, (256 / ncp_) / 8 * 8), cdiv(mt_, 8) * 8);
            hipLaunchKernelGGL(fn, dim3(ns_ * ncp_), dim3(256), BR_LDS, st, (const h16_t*)A, lda, (const h16_t*)B, ldb, (h16_t*)C, ldc, bias, M, N, ncp_, ns_,
                               cdiv(mt_, ns_));
            MNN_LAUNCH_CHECK();
            return MNN_OK;                                         // (never touches raised_: the product slot keeps its own LDS-limit bookkeeping)
        }
    }
#endif
    static bool raised_[64][4];
    int dev = 0;
    MNN_HIP(hipGetDevice(&dev));
    MNN_REQUIRE(dev >= 0 && dev < 64, "gemm_bres: device index %d", dev);
    const int vi = (K == 448 ? 0 : 2) + (f16 ? 1 : 0);
    if (!raised_[dev][vi]) {
        MNN_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, BR_LDS));
        raised_[dev][vi] = true;
    }
    const int ncp = cdiv(N, 256);
    const int mt = M / 128;
    int nslab = std::max(8, (256 / ncp) / 8 * 8);                  // one workgroup per CU, slab count a multiple of 8 (XCD mapping)
    nslab = std::min(nslab, cdiv(mt, 8) * 8);
    const int tiles_per_slab = cdiv(mt, nslab);
    hipLaunchKernelGGL(fn, dim3(nslab * ncp), dim3(256), BR_LDS, st, (const h16_t*)A, lda, (const h16_t*)B, ldb, (h16_t*)C, ldc, bias, M, N, ncp, nslab,
                       tiles_per_slab);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
