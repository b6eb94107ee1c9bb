// CU-resident LSTM recurrence (gfx950): one workgroup owns FOUR batch rows of a layer for the whole sequence and holds the layer's ENTIRE
// recurrent matrix on its CU -- three quarters of it in the four waves' registers (AGPRs + VGPRs: 384 of a lane's 512 at one wave per
// SIMD), the last quarter in LDS.  Nothing is handed between workgroups: the step is  h[t-1] (4 x u, LDS) -> 128 MFMAs per wave against
// register / LDS weights -> gate pointwise -> h[t] (LDS) -> one workgroup barrier.  The row-parallel form (lstm_rowpar.hip) spends
// 4.3 us per timestep of a 256-unit layer, most of it on the L2 round trips of its hand-offs (tile stores -> flag -> poll -> tile loads);
// here the chain is the MFMA stream itself (16 weight tiles x 8 k-steps of v_mfma_f32_16x16x32 = 2048 cycles) plus the pointwise.
//
// Matrix-core layout (v_mfma_f32_16x16x32_{f16,bf16}; D[m][n] += A[m][k] B[k][n]; lane l holds A[l & 15][8 (l >> 4) ..+7], B[8 (l >> 4) ..+7][l & 15],
// D[4 (l >> 4) + r][l & 15], r = 0..3):
//   A = weights: row m = 4 * (unit in tile) + gate  (tile = 4 units x {i, g, f, o}), so a lane's four D registers are the four gates of ONE unit;
//   B = h[t-1]: column n = batch row n & 3 -- the four rows are REPLICATED over the four 4-lane banks of a 16-lane group (every lane reads
//       row l & 3), so every bank of D holds the same 4 rows and nothing has to be zero-padded;
//   four tiles (16 units) form a group: bank b of the lanes keeps tile b's result (three v_cndmask per register), and then every lane owns
//   exactly one (row, unit) pair of the group: row l & 3, unit 64 w + 16 q + 4 ((l & 15) >> 2) + (l >> 4).
// A wave owns 64 units = 4 groups = 4 pairs per lane; the 12 of 16 MFMA columns that repeat rows are the price of a 4-row workgroup (the
// matrix must be streamed through the matrix cores once per step whatever the row count: 2048 cycles), which in turn is what spreads the
// pointwise of B x u pairs over all 1024 SIMDs.
#include "common.h"

__device__ __forceinline__ float dpp_xor2(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_xor1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)); }
#define RES_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#ifdef RES_TRACE     // development only (profiles/tools/resident_trace.py): wall-clock stamps (100 MHz) of wave 0 of workgroup 0, [direction][step][stage]
__device__ long long res_trace[2][512][12];
#define RES_TR(k) do { tr_[k] = wall_clock64(); } while (0)
#define RES_TR_DECL long long tr_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define RES_TR_FLUSH(dir, step) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 512) { _Pragma("unroll") for (int k_ = 0; k_ < 12; ++k_) res_trace[dir][step][k_] = tr_[k_]; } } while (0)
extern "C" int mnn_lstm_resident_trace(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(res_trace), sizeof(res_trace)) == hipSuccess ? 0 : -2;
}
#else
#define RES_TR(k) do { } while (0)
#define RES_TR_DECL do { } while (0)
#define RES_TR_FLUSH(dir, step) do { } while (0)
#endif

struct ResFwdArgs {
    const h16_t* xproj; const h16_t* wh_t; h16_t* gates; float* c; h16_t* h; h16_t* y; const uint8_t* mask;
    h16_t* hT; int ld_hT; h16_t* yT; int ld_yT;
    int T, B; float kp;
};
// Several independent layers of the same shape in ONE launch (mnn_lstm_resident_fwd_multi: the per-track generators of the jamming mode,
// multinn_jamming.py:40-68): a workgroup owns four rows of ONE job for the whole sequence and hands nothing to another workgroup, so the jobs'
// row groups simply share the grid (any number of rounds).  gridDim.x = njobs * B / 4; job = blockIdx.x / (B / 4).
#define RES_MAX_JOBS 8
struct ResFwdJobs { ResFwdArgs job[RES_MAX_JOBS]; int njobs; };

template <int U> struct ResGeom {
    static_assert(U == 256, "the CU-resident recurrence is sized for 256-unit layers (512 KB of 16-bit recurrent weights)");
    static constexpr int UW = U / 4;            // units per wave
    static constexpr int NG = UW / 16;          // groups of four tiles (16 units)
    static constexpr int NT = UW / 4;           // weight tiles (4 units x 4 gates) per wave
    static constexpr int KS = U / 32;           // k-steps of 32
    static constexpr int KR = 6;                // k-steps whose weights stay in registers
    static constexpr int KL = KS - KR;          // ... and in LDS
    static constexpr int PH = U * 2 + 64;       // pitch of a state row in LDS: rows 16 banks apart (the 4 rows x 4 k-groups of a B read cover all 64)
    static constexpr int PX = U * 8 + 128;      // pitch of an xproj row in LDS (rows 32 banks apart)
    static constexpr int OFF_W = 0;
    static constexpr int WPW = NT * KL;         // LDS fragments (1 KB each) per wave
    static constexpr int OFF_H = OFF_W + 4 * WPW * 1024;
    static constexpr int OFF_Y = OFF_H + 2 * 4 * PH;
    static constexpr int OFF_X = OFF_Y + 2 * 4 * PH;
    static constexpr int OFF_M = OFF_X + 2 * 4 * PX;
    static constexpr int LDS = OFF_M + 2 * 4 * U;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

// row group of a workgroup: workgroups are dealt round-robin over the 8 XCDs, so XCD x takes the contiguous rows [x nblk/8, (x+1) nblk/8) * 4 --
// the 8-byte pieces that neighbouring row groups write into one 128-byte line of a transposed output then meet in ONE L2
__device__ __forceinline__ int res_row_group(int b, int nblk) { return (nblk & 7) == 0 ? (b & 7) * (nblk >> 3) + (b >> 3) : b; }

typedef __attribute__((address_space(1))) const void* res_gptr_t;
typedef __attribute__((address_space(3))) void* res_lptr_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define RES_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void res_wait_all_but(int n) {       // loads, stores and LDS-DMA of a wave complete in issue order (MI355X_MICROARCH.md)
    switch (n) {
        RES_VMC(0) RES_VMC(1) RES_VMC(2) RES_VMC(3) RES_VMC(4) RES_VMC(5) RES_VMC(6) RES_VMC(7) RES_VMC(8) RES_VMC(9) RES_VMC(10) RES_VMC(11) RES_VMC(12)
        RES_VMC(13) RES_VMC(14) RES_VMC(15) RES_VMC(16)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int U, typename F, bool DROP, bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_res_fwd_kernel(ResFwdJobs J) {
    typedef ResGeom<U> G;
    typedef typename F::x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 3, bank = (lane & 15) >> 2, g4 = lane >> 4;
    const int nblk = (int)gridDim.x / J.njobs, jid = (int)blockIdx.x / nblk;   // this workgroup's job (uniform) and its row group inside it
    const ResFwdArgs A = J.job[jid];
    const int T = A.T, B = A.B;
    const int row0 = 4 * res_row_group((int)blockIdx.x - jid * nblk, nblk);
    const size_t us = (size_t)B * U;
    const float ikp = 1.0f / A.kp;

    // ---- the recurrent matrix: A fragments of this wave's 16 tiles, k-steps < KR in registers, the rest in LDS ----
    frag_t wr[G::NT][G::KR];
    {
        const int m = lane & 15;
        uint4* wl = reinterpret_cast<uint4*>(smem + G::OFF_W);
#pragma unroll
        for (int tl = 0; tl < G::NT; ++tl) {
            const int unit = G::UW * w + 4 * tl + (m >> 2);
            const h16_t* src = A.wh_t + (size_t)gate_perm_col(m & 3, unit) * U + 8 * g4;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + 32 * s);
                if (s < G::KR) {
                    wr[tl][s] = __builtin_bit_cast(frag_t, v);
                } else wl[(w * G::WPW + tl * G::KL + (s - G::KR)) * 64 + lane] = v;
            }
        }
        // k-steps 0..3 are pinned to AGPRs, which the matrix cores read directly (left to itself the register allocator treats AGPRs as spill
        // space and copies every fragment back with four v_accvgpr_read per MFMA); 4..5 stay in VGPRs.  Pinned only after ALL loads have been
        // issued: a pin right behind its load makes every load wait for the one before (64 memory round trips at the start of every launch)
#pragma unroll
        for (int tl = 0; tl < G::NT; ++tl)
#pragma unroll
            for (int s = 0; s < 4; ++s) asm volatile("" : "+a"(wr[tl][s]));
    }
    // zero state: h[-1] = 0 in the first state buffer
    for (int i = tid; i < 4 * G::PH / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::OFF_H)[i] = 0u;

    // ---- staging of a step's xproj rows (4 x 8u bytes) and keep bytes (4 x u) a step ahead, global -> LDS without registers: wave w moves row w ----
    const char* xsrc = reinterpret_cast<const char*>(A.xproj) + ((size_t)(row0 + w) * U) * 8 + lane * 16;
    const uint8_t* msrc = DROP ? A.mask + (size_t)(row0 + w) * U + lane * 4 : nullptr;
    // LDS-DMA in inline assembly: the compiler puts `s_waitcnt vmcnt(0)` in front of every LDS read that may alias the destination of a
    // __builtin_amdgcn_global_load_lds -- here the reads of THIS step's staged rows, one buffer over: the whole memory latency on the chain of
    // every step (3.5 us per step measured).  The step waits for its own DMA itself (res_wait_all_but, in-order completion), one barrier later.
    auto dma16 = [&](const void* g, const void* l) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"((unsigned)(uintptr_t)(res_lptr_t)const_cast<void*>(l)), "v"(g) : "memory", "m0");
    };
    auto dma4 = [&](const void* g, const void* l) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"((unsigned)(uintptr_t)(res_lptr_t)const_cast<void*>(l)), "v"(g) : "memory", "m0");
    };
    auto stage = [&](int t, int buf) {
        const char* xb = xsrc + (size_t)t * us * 8;
        char* xs = smem + G::OFF_X + buf * 4 * G::PX + w * G::PX;
#pragma unroll
        for (int i = 0; i < U * 8 / 1024; ++i) dma16(xb + i * 1024, xs + i * 1024);
        if (DROP) dma4(msrc + (size_t)t * us, smem + G::OFF_M + buf * 4 * U + w * U);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RES_BARRIER();

    // ---- what leaves a step through LDS: h / y row-major (16-byte pieces) and h^T / y^T (a unit's four rows = 8 bytes).  Branch-free: buffer
    // stores whose offset is pushed out of range are dropped by the hardware, so the step's code stays one scheduling region ----
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)A.h, 0, (int)min((size_t)T * us * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(DROP ? (void*)A.y : (void*)A.h, 0, (int)min((size_t)T * us * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hT = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)A.hT : (void*)A.h, 0, SAVE ? (int)min((size_t)U * A.ld_hT * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_yT = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)A.yT : (void*)A.h, 0, SAVE ? (int)min((size_t)U * A.ld_yT * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const int er = (tid >> 5) & 3, epc = tid & 31;
    const unsigned vo_row = (unsigned)(er * U + epc * 8) * 2u;                  // row-major piece of this thread inside the step's 4 x u block
    const unsigned vo_lo = tid < 128 ? vo_row : OOB, vo_hi = tid >= 128 ? vo_row : OOB;
    const unsigned vo_hT = (unsigned)tid * (unsigned)A.ld_hT * 2u, vo_yT = (unsigned)tid * (unsigned)A.ld_yT * 2u;
    // (the SCALAR offset of a buffer access is not range-checked: only the lane offset can push a store out of range)
    // Two halves: the LDS reads are issued early in the first group's MFMA stream and consumed (packed, stored) a few k-steps later, so that
    // their latency is under the MFMAs (read and used in one place they stalled the stream for ~1000 cycles per step).
    u32x4_t e_hrow, e_yrow;
    unsigned e_col[2][4];
    auto emit_read = [&](int t) {                  // outputs of step t: h[t] sits in state buffer (t + 1) & 1, y[t] in y buffer t & 1
        const char* hb = smem + G::OFF_H + ((t + 1) & 1) * 4 * G::PH;
        const char* yb = smem + G::OFF_Y + (t & 1) * 4 * G::PH;
        e_hrow = *reinterpret_cast<const u32x4_t*>(hb + er * G::PH + epc * 16);
        if (DROP) e_yrow = *reinterpret_cast<const u32x4_t*>(yb + er * G::PH + epc * 16);
        if (SAVE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e_col[0][r] = reinterpret_cast<const h16_t*>(hb)[tid + r * (G::PH / 2)];
                if (DROP) e_col[1][r] = reinterpret_cast<const h16_t*>(yb)[tid + r * (G::PH / 2)];
            }
        }
    };
    auto emit_store = [&](int t) {                 // t = -1: nothing leaves
        const int tc = t < 0 ? 0 : t;
        const unsigned none = t < 0 ? OOB : 0u;                                  // wave-uniform kill bits, OR-ed into the lane offsets
        const unsigned so_row = (unsigned)(((size_t)tc * us + (size_t)row0 * U) * 2);
        if (DROP) {
            __builtin_amdgcn_raw_buffer_store_b128(e_yrow, rs_y, vo_hi | none, so_row, 0);
            __builtin_amdgcn_raw_buffer_store_b128(e_hrow, rs_h, vo_lo | (t + 1 == T ? 0u : OOB), so_row, 0);   // only the final state reads h
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(e_hrow, rs_h, vo_lo | none, so_row, 0);
        }
        if (SAVE) {
            u32x2_t hc, yc;
            hc[0] = e_col[0][0] | (e_col[0][1] << 16); hc[1] = e_col[0][2] | (e_col[0][3] << 16);
            if (DROP) { yc[0] = e_col[1][0] | (e_col[1][1] << 16); yc[1] = e_col[1][2] | (e_col[1][3] << 16); }
            else yc = hc;
            const int tn = t + 1 < T ? t + 1 : 0;
            __builtin_amdgcn_raw_buffer_store_b64(hc, rs_hT, vo_hT | ((t < 0 || t + 1 >= T) ? OOB : 0u), (unsigned)(((size_t)tn * B + row0) * 2), 0);
            __builtin_amdgcn_raw_buffer_store_b64(yc, rs_yT, vo_yT | none, (unsigned)(((size_t)tc * B + row0) * 2), 0);
        }
    };
    constexpr int EMIT_STORES = (DROP ? 2 : 1) + (SAVE ? 2 : 0);

    float creg[G::NG];
#pragma unroll
    for (int q = 0; q < G::NG; ++q) creg[q] = 0.f;
    const uint4* wl = reinterpret_cast<const uint4*>(smem + G::OFF_W) + (size_t)w * G::WPW * 64 + lane;
    // order of the k-steps inside a group: the two LDS-resident ones sit two register k-steps apart, so that the four fragment registers
    // of the first can be re-requested for the second while 8 MFMAs run
    constexpr int ORD[8] = {0, 1, 6, 2, 3, 4, 7, 5};
    constexpr int PFB = 3;                          // state fragments requested ahead of their MFMAs (LDS latency ~ 2 k-steps of 4 MFMAs)

    static_assert(G::KL == 2, "the fragment ring below assumes two LDS-resident k-steps per tile");
    frag_t lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) lw[j] = __builtin_bit_cast(frag_t, wl[(j * G::KL + 0) * 64]);
    RES_TR_DECL;
    for (int t = 0; t < T; ++t) {
        RES_TR(0);
#ifdef RES_TRACE
        tr_[9] = clock64();
#endif
        stage(t + 1 < T ? t + 1 : t, (t + 1) & 1);              // unconditional (clamped); everything below is younger than these requests
        asm volatile("" ::: "memory");
        const char* hin = smem + G::OFF_H + (t & 1) * 4 * G::PH + row * G::PH + g4 * 16;
        char* hout = smem + G::OFF_H + ((t + 1) & 1) * 4 * G::PH + row * G::PH;
        char* yout = smem + G::OFF_Y + (t & 1) * 4 * G::PH + row * G::PH;
        const char* xin = smem + G::OFF_X + (t & 1) * 4 * G::PX + row * G::PX;
        const uint8_t* min_ = reinterpret_cast<const uint8_t*>(smem + G::OFF_M + (t & 1) * 4 * U + row * U);
        // the step's 4 x u blocks of the saved gates / c as buffer resources (scalar base, one 32-bit lane offset)
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)(A.gates + ((size_t)t * us + (size_t)row0 * U) * 4) : (void*)A.c, 0, 4 * U * 8, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)(A.c + (size_t)t * us + (size_t)row0 * U), 0, 4 * U * 4, 0x00020000);
        mnn_f32x4 acc[2][4];
        // the MFMA stream of one group of four tiles; `filler(i)` is called between k-steps (the pointwise of the previous group)
        auto group = [&](int q, mnn_f32x4 (&ac)[4]) {
            // the group's two LDS-resident k-steps sit at positions 2 and 6 of ORD.  `lw` arrives holding the first (requested while the previous
            // group ran its last k-steps -- across the timestep boundary too: weights do not change), is re-requested for the second right behind
            // the first's MFMAs (four k-steps ahead) and for the NEXT group's first right behind the second's.
            frag_t b[G::KS];
#pragma unroll
            for (int i = 0; i < PFB; ++i) b[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 64 * ORD[i]));
#pragma unroll
            for (int i = 0; i < G::KS; ++i) {
                const int s = ORD[i];
                if (i + PFB < G::KS) b[i + PFB] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 64 * ORD[i + PFB]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const mnn_f32x4 c0 = i == 0 ? mnn_f32x4{0.f, 0.f, 0.f, 0.f} : ac[j];
                    ac[j] = F::mfma16(s >= G::KR ? lw[j] : wr[4 * q + j][s < G::KR ? s : 0], b[i], c0);
                }
                if (s >= G::KR) {
                    const int qn = s + 1 < G::KS ? q : (q + 1) % G::NG, sn = s + 1 < G::KS ? s + 1 : G::KR;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lw[j] = __builtin_bit_cast(frag_t, wl[((4 * qn + j) * G::KL + (sn - G::KR)) * 64]);
                }
                if (q == 0 && i == 0) emit_read(t - 1);         // the previous step's outputs leave in the shadow of this step's first MFMAs
                if (q == 0 && i == 4) emit_store(t - 1);
                // the state fragments stay PFB k-steps ahead (left alone, the scheduler requests all eight at the top of the group: 16 registers
                // more than there are); vector / scalar / transcendental work, stores and LDS writes of the previous group's pointwise may cross
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x40 | 0x200);
            }
        };
        uint2 xv;
        unsigned mkb = 0u;
        auto pointwise_pre = [&](int q) {                  // the group's LDS operands, requested in front of the next group's MFMA stream
            const int unit = G::UW * w + 16 * q + 4 * bank + g4;
            xv = *reinterpret_cast<const uint2*>(xin + unit * 8);
            if (DROP) mkb = min_[unit];
        };
        auto pointwise = [&](int q, const mnn_f32x4 (&ac)[4]) {
            // bank b keeps tile b: every lane owns one (row, unit) pair of the group
            const int unit = G::UW * w + 16 * q + 4 * bank + g4;
            float z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float lo = bank & 1 ? ac[1][r] : ac[0][r], hi = bank & 1 ? ac[3][r] : ac[2][r];
                z[r] = bank & 2 ? hi : lo;
            }
            const float gi = fast_sigmoid(z[0] + F::lo(xv.x)), gg = fast_tanh(z[1] + F::hi(xv.x));
            const float gf = fast_sigmoid(z[2] + F::lo(xv.y)), go = fast_sigmoid(z[3] + F::hi(xv.y));
            const float cv = gg * gi + creg[q] * gf;
            const float hv = fast_tanh(cv) * go;
            creg[q] = cv;
            const h16_t hb = F::cvt(hv);
            reinterpret_cast<h16_t*>(hout)[unit] = hb;
            if (DROP) reinterpret_cast<h16_t*>(yout)[unit] = F::cvt(F::f32(hb) * ikp * (float)mkb);
            const int e = row * U + unit;                                        // inside the step's 4 x u block
            if (SAVE) {
                u32x2_t pk;
                pk[0] = pack2<F>(gi, gg);
                pk[1] = pack2<F>(gf, go);
                __builtin_amdgcn_raw_buffer_store_b64(pk, rs_g, e * 8, 0, 0);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(cv), rs_c, e * 4, 0, 0);
        };
        RES_TR(1);
        group(0, acc[0]);
        RES_TR(2);
#pragma unroll
        for (int q = 0; q < G::NG; ++q) {
            pointwise_pre(q);
            if (q + 1 < G::NG) group(q + 1, acc[(q + 1) & 1]);
            pointwise(q, acc[q & 1]);
            RES_TR(3 + q);
        }
        res_wait_all_but(EMIT_STORES + G::NG * (SAVE ? 2 : 1));               // the staged rows of step t + 1 are in LDS (the pointwise stores behind them may not be out)
        RES_TR(7);
        RES_BARRIER();
        RES_TR(8);
        RES_TR_FLUSH(0, t);
    }
    emit_read(T - 1);
    emit_store(T - 1);
}

// ------------------------------------------------------------------------------------------------------------------
// backward:  dh = dh_ext[t] (/ kp * keep when a mask is given) + dz[t+1] . Wh ;  gate backward -> dz[t], dc   (rnn.py:124 LSTMBlockCell autodiff)
// Same ownership (four rows per workgroup, the whole [u, 4u] matrix on the CU).  A = Wh rows (16 output units x 32 gate columns per MFMA),
// B = dz[t+1] (4 rows, replicated over the banks), K = 4u = 32 k-steps: a wave's four accumulators run over all of them, bank b keeps tile b,
// and a lane then owns FOUR CONSECUTIVE units of one row: u0 = 64 w + 16 bank + 4 (l >> 4), row l & 3.  Its operands (saved gates 32 bytes,
// c[t-1] 16, dh_ext 16, keep bytes 4) are five contiguous loads issued a step ahead.  The step is serial by nature -- all of dz[t+1] is needed
// before the first MFMA and all MFMAs before the pointwise -- so the outputs of the previous step (dz row-major and dz^T) leave from the
// LDS tile in the shadow of the MFMA phase.
// ------------------------------------------------------------------------------------------------------------------
struct ResBwdArgs {
    const float* dh_ext; const h16_t* wh_p; const h16_t* gates; const float* c; const uint8_t* mask;
    h16_t* dzc; h16_t* dzT; int ld_t; float* db_p;
    int T, B; float kp;
};
struct ResBwdJobs { ResBwdArgs job[RES_MAX_JOBS]; int njobs; };

template <int U> struct ResBwdGeom {
    static_assert(U == 256, "the CU-resident recurrence is sized for 256-unit layers");
    static constexpr int KS = 4 * U / 32;       // k-steps of 32 over the gate columns
    static constexpr int KL = KS / 4;           // every fourth k-step's fragments live in LDS
    static constexpr int PZ = 4 * U * 2 + 64;   // pitch of a dz row in LDS (rows 16 banks apart)
    static constexpr int OFF_W = 0;
    static constexpr int OFF_Z = OFF_W + 4 * 4 * KL * 1024;
    static constexpr int LDS = OFF_Z + 2 * 4 * PZ;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int U, typename F, bool DROP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_res_bwd_kernel(ResBwdJobs J) {
    typedef ResBwdGeom<U> G;
    typedef typename F::x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 3, bank = (lane & 15) >> 2, g4 = lane >> 4;
    const int nblk = (int)gridDim.x / J.njobs, jid = (int)blockIdx.x / nblk;   // this workgroup's job (uniform) and its row group inside it
    const ResBwdArgs A = J.job[jid];
    const int T = A.T, B = A.B;
    const int row0 = 4 * res_row_group((int)blockIdx.x - jid * nblk, nblk);
    const size_t us = (size_t)B * U;
    const float ikp = 1.0f / A.kp;
    const int u0 = 64 * w + 16 * bank + 4 * g4;                              // this lane's four units (after the bank selection)

    // ---- Wh [u, 4u]: A fragments of this wave's four 16-unit tiles; k-steps s % 4 != 3 in registers (0..15 of them pinned to AGPRs), s % 4 == 3 in LDS ----
    frag_t wr[4][G::KS - G::KL];
    {
        uint4* wl = reinterpret_cast<uint4*>(smem + G::OFF_W);
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const h16_t* src = A.wh_p + (size_t)(64 * w + 16 * tl + (lane & 15)) * (4 * U) + 8 * g4;
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(src + 32 * s);
                if ((s & 3) == 3) wl[((w * 4 + tl) * G::KL + (s >> 2)) * 64 + lane] = v;
                else {
                    const int si = s - (s >> 2);
                    wr[tl][si] = __builtin_bit_cast(frag_t, v);
                }
            }
        }
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int si = 0; si < 16; ++si) asm volatile("" : "+a"(wr[tl][si]));     // (pinned after all loads have been issued: see the forward)
    }
    for (int i = tid; i < 4 * G::PZ / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::OFF_Z)[i] = 0u;     // dz[T] = 0

    // ---- the step's operands: buffer loads with a scalar per-step base and one lane offset (units u0 .. u0+3 of row `row`) ----
    const int lim = (int)min((size_t)T * us * 8, (size_t)0x7fffffff);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)A.gates, 0, lim, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)A.c, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)A.dh_ext, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(DROP ? (void*)A.mask : (void*)A.c, 0, (int)min((size_t)T * us, (size_t)0x7fffffff), 0x00020000);
    const unsigned vo_e = (unsigned)(row * U + u0);                          // element offset inside a step's 4 x u block
    u32x4_t gq0, gq1, cq, dq;
    unsigned mq = 0u;
    auto request = [&](int t) {                                              // operands of step t (t >= 0)
        const unsigned so = (unsigned)(((size_t)t * us + (size_t)row0 * U));
        gq0 = __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo_e * 8, so * 8, 0);
        gq1 = __builtin_amdgcn_raw_buffer_load_b128(rs_g, vo_e * 8 + 16, so * 8, 0);
        const unsigned sp = (unsigned)(((size_t)(t > 0 ? t - 1 : 0) * us + (size_t)row0 * U));
        cq = __builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_e * 4, sp * 4, 0);       // c[t-1] (t = 0: read and ignored)
        dq = __builtin_amdgcn_raw_buffer_load_b128(rs_d, vo_e * 4, so * 4, 0);
        if (DROP) mq = __builtin_amdgcn_raw_buffer_load_b32(rs_m, vo_e, so, 0);
    };
    float cnext[4], dcreg[4], dbv[4][4];
    {
        const u32x4_t c_last = __builtin_amdgcn_raw_buffer_load_b128(rs_c, vo_e * 4, (unsigned)(((size_t)(T - 1) * us + (size_t)row0 * U)) * 4, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cnext[r] = __uint_as_float(c_last[r]);
            dcreg[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) dbv[g][r] = 0.f;
        }
    }
    request(T - 1);

    // ---- what leaves through the LDS tile: dz row-major [t][row][4u] (16-byte pieces) and dz^T (a column's four rows = 8 bytes) ----
    constexpr unsigned OOB = 0x80000000u;
    const size_t N = (size_t)T * B;
    const bool kb = A.ld_t == 0;
    const __amdgpu_buffer_rsrc_t rs_zc = __builtin_amdgcn_make_buffer_rsrc(A.dzc ? (void*)A.dzc : (void*)A.c, 0, A.dzc ? (int)min(N * 4 * U * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_zt = __builtin_amdgcn_make_buffer_rsrc(A.dzT ? (void*)A.dzT : (void*)A.c, 0,
                                                                           A.dzT ? (int)min(kb ? N * 4 * U * 2 : (size_t)4 * U * A.ld_t * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const unsigned vo_zc = (unsigned)((tid >> 6) * 4 * U + (tid & 63) * 8) * 2u;
    const unsigned vo_zt = kb ? (unsigned)(tid * 32 + (row0 & 31)) * 2u : (unsigned)tid * (unsigned)A.ld_t * 2u;
    const unsigned st_zt = kb ? 256u * 64u : 256u * (unsigned)A.ld_t * 2u;                      // between this thread's four columns
    // (the SCALAR offset of a buffer access is not range-checked: only the lane offset can push a store out of range)
    // Two halves (as in the forward): LDS reads early in the MFMA stream, packing and stores a few k-steps later.
    // The four columns go in two rounds (8 + 8 registers instead of 24 at once: the wave has none to spare).
    u32x4_t e_row;
    unsigned e_col[2][4];
    auto emit_read = [&](int buf, int half) {
        const char* zb = smem + G::OFF_Z + buf * 4 * G::PZ;
        e_row = *reinterpret_cast<const u32x4_t*>(zb + (tid >> 6) * G::PZ + ((tid & 63) + 64 * half) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) e_col[i][r] = reinterpret_cast<const h16_t*>(zb)[tid + 256 * (2 * half + i) + r * (G::PZ / 2)];
    };
    auto emit_store = [&](int tt, int half) {       // dz[tt] (tt >= T: nothing)
        const unsigned none = tt >= T ? OOB : 0u;
        const int tc = tt >= T ? 0 : tt;
        const unsigned so_c = (unsigned)(((size_t)tc * B + row0) * 4 * U * 2);
        __builtin_amdgcn_raw_buffer_store_b128(e_row, rs_zc, (vo_zc + 64 * 16 * half) | none, so_c, 0);
        const unsigned so_t = kb ? (unsigned)((((size_t)tc * (B >> 5) + (row0 >> 5)) * 4 * U) * 64) : (unsigned)(((size_t)tc * B + row0) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x2_t v;
            v[0] = e_col[i][0] | (e_col[i][1] << 16);
            v[1] = e_col[i][2] | (e_col[i][3] << 16);
            __builtin_amdgcn_raw_buffer_store_b64(v, rs_zt, vo_zt | none, so_t + st_zt * (2 * half + i), 0);
        }
    };

    const uint4* wl = reinterpret_cast<const uint4*>(smem + G::OFF_W) + (size_t)w * 4 * G::KL * 64 + lane;
    constexpr int PFB = 3;
    RES_BARRIER();
    RES_TR_DECL;
    for (int kk = 0; kk < T; ++kk) {
        const int t = T - 1 - kk;
        RES_TR(0);
        const char* zin = smem + G::OFF_Z + (kk & 1) * 4 * G::PZ + row * G::PZ + g4 * 16;
        char* zout = smem + G::OFF_Z + ((kk + 1) & 1) * 4 * G::PZ + row * G::PZ;
        mnn_f32x4 acc[4];
        {
            frag_t b[G::KS], lw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) lw[j] = __builtin_bit_cast(frag_t, wl[(j * G::KL + 0) * 64]);
#pragma unroll
            for (int i = 0; i < PFB; ++i) b[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(zin + 64 * i));
#pragma unroll
            for (int s = 0; s < G::KS; ++s) {
                if (s + PFB < G::KS) b[s + PFB] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(zin + 64 * (s + PFB)));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const mnn_f32x4 c0 = s == 0 ? mnn_f32x4{0.f, 0.f, 0.f, 0.f} : acc[j];
                    acc[j] = F::mfma16((s & 3) == 3 ? lw[j] : wr[j][s - (s >> 2)], b[s], c0);
                }
                if ((s & 3) == 3 && s + 4 < G::KS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) lw[j] = __builtin_bit_cast(frag_t, wl[(j * G::KL + (s >> 2) + 1) * 64]);
                }
                if (s == 0) emit_read(kk & 1, 0);               // the previous step's outputs, in the shadow of the MFMA stream
                if (s == 5) emit_store(t + 1, 0);
                if (s == 6) emit_read(kk & 1, 1);
                if (s == 11) emit_store(t + 1, 1);
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x40 | 0x200);
            }
        }
        RES_TR(1);
        // ---- pointwise: bank b keeps tile b; register r = unit u0 + r ----
        float dhr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = bank & 1 ? acc[1][r] : acc[0][r], hi = bank & 1 ? acc[3][r] : acc[2][r];
            dhr[r] = bank & 2 ? hi : lo;
        }
        h16_t b4[4][4];                                 // [gate][unit]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned g01 = r < 2 ? gq0[2 * r] : gq1[2 * r - 4], g23 = r < 2 ? gq0[2 * r + 1] : gq1[2 * r - 3];
            const float gi = F::lo(g01), gg = F::hi(g01), gf = F::lo(g23), go = F::hi(g23);
            const float dv = __uint_as_float(dq[r]);
            const float dh = (DROP ? dv * ikp * (float)((mq >> (8 * r)) & 0xffu) : dv) + dhr[r];
            const float tc = fast_tanh(cnext[r]);
            const float d_o = dh * tc;
            const float d_c = dh * go * (1.f - tc * tc) + dcreg[r];
            const float cprev = t > 0 ? __uint_as_float(cq[r]) : 0.f;
            const float dzv[4] = {d_c * gg * gi * (1.f - gi), d_c * gi * (1.f - gg * gg), d_c * cprev * gf * (1.f - gf), d_o * go * (1.f - go)};
            dcreg[r] = d_c * gf;
            cnext[r] = cprev;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b4[g][r] = F::cvt(dzv[g]);
                dbv[g][r] += F::f32(b4[g][r]);          // the (16-bit) values the weight-gradient GEMMs see
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2_t v;
            v[0] = (unsigned)b4[g][0] | ((unsigned)b4[g][1] << 16);
            v[1] = (unsigned)b4[g][2] | ((unsigned)b4[g][3] << 16);
            *reinterpret_cast<u32x2_t*>(zout + gate_perm_col(g, u0) * 2) = v;
        }
        RES_TR(2);
        request(t > 0 ? t - 1 : 0);                     // unconditional (clamped): behind a condition the compiler copies the freshly requested registers at the join and waits for the loads to do it
        RES_BARRIER();
        RES_TR(3);
        RES_TR_FLUSH(1, kk);
    }
    emit_read(T & 1, 0);
    emit_store(0, 0);
    emit_read(T & 1, 1);
    emit_store(0, 1);
    if (A.db_p != nullptr) {                            // bias gradient: sums over this workgroup's four rows and all steps
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = dbv[g][r];
                v += dpp_xor1(v);
                v += dpp_xor2(v);
                if (row == 0) atomicAdd(A.db_p + gate_perm_col(g, u0 + r), v);
            }
    }
}

static hipError_t res_prepare();
// (also raises the kernels' dynamic-LDS limit on this device: the host asks before every use, so never for the first time under stream capture)
extern "C" int mnn_lstm_resident_ok(int B, int units) {
    if (!(units == 256 && B > 0 && (B & 3) == 0)) return 0;
    return res_prepare() == hipSuccess ? 1 : 0;
}
static bool res_shape_ok(int B, int units) { return units == 256 && B > 0 && (B & 3) == 0; }

typedef void (*res_fwd_fn)(ResFwdJobs);
template <typename F> static res_fwd_fn res_fwd_pick(bool drop, bool save) {
    if (drop) return save ? lstm_res_fwd_kernel<256, F, true, true> : lstm_res_fwd_kernel<256, F, true, false>;
    return save ? lstm_res_fwd_kernel<256, F, false, true> : lstm_res_fwd_kernel<256, F, false, false>;
}
static res_fwd_fn res_fwd_kernel(bool f16, bool drop, bool save) { return f16 ? res_fwd_pick<Fp16F>(drop, save) : res_fwd_pick<Bf16F>(drop, save); }
typedef void (*res_bwd_fn)(ResBwdJobs);
static res_bwd_fn res_bwd_kernel(bool f16, bool drop) {
    if (f16) return drop ? lstm_res_bwd_kernel<256, Fp16F, true> : lstm_res_bwd_kernel<256, Fp16F, false>;
    return drop ? lstm_res_bwd_kernel<256, Bf16F, true> : lstm_res_bwd_kernel<256, Bf16F, false>;
}
static hipError_t res_prepare() {
    static bool done[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    for (int i = 0; i < 8; ++i) {
        e = hipFuncSetAttribute((const void*)res_fwd_kernel(i & 1, i & 2, i & 4), hipFuncAttributeMaxDynamicSharedMemorySize, ResGeom<256>::LDS);
        if (e != hipSuccess) return e;
    }
    for (int i = 0; i < 4; ++i) {
        e = hipFuncSetAttribute((const void*)res_bwd_kernel(i & 1, i & 2), hipFuncAttributeMaxDynamicSharedMemorySize, ResBwdGeom<256>::LDS);
        if (e != hipSuccess) return e;
    }
    done[dev] = true;
    return hipSuccess;
}

static int res_fwd_fill(const mnn_lstm_fwd_layer* L, int T, int B, float keep_prob, ResFwdArgs& a) {
    MNN_REQUIRE(L && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_resident_fwd: bad arguments");
    MNN_REQUIRE(res_shape_ok(B, L->units), "mnn_lstm_resident_fwd: units must be 256 and B a multiple of 4 (B=%d u=%d)", B, L->units);
    MNN_REQUIRE(L->xproj && L->wh_t && L->c && L->h, "mnn_lstm_resident_fwd: null pointer");
    MNN_REQUIRE(L->xproj_bf16 != 0, "mnn_lstm_resident_fwd: the input projection must be in the layer's 16-bit type (gate-minor, bias included)");
    MNN_REQUIRE(L->h0 == nullptr && L->c0 == nullptr, "mnn_lstm_resident_fwd: an initial state is not supported by this form (zero state per window)");
    MNN_REQUIRE(L->hT == nullptr || (L->ld_hT >= T * B && (L->ld_hT & 3) == 0), "mnn_lstm_resident_fwd: ld_hT too small / not a multiple of 4");
    MNN_REQUIRE(L->yT == nullptr || (L->ld_yT >= T * B && (L->ld_yT & 3) == 0), "mnn_lstm_resident_fwd: ld_yT too small / not a multiple of 4");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                "mnn_lstm_resident_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    MNN_REQUIRE((L->gates != nullptr) == (L->hT != nullptr) && (L->gates != nullptr) == (L->yT != nullptr),
                "mnn_lstm_resident_fwd: the saved gates, hT and yT come together (training) or not at all");
    MNN_REQUIRE((size_t)T * B * 256 * 8 < ((size_t)1 << 31) && (size_t)256 * (size_t)(L->ld_hT > L->ld_yT ? L->ld_hT : L->ld_yT) * 2 < ((size_t)1 << 31),
                "mnn_lstm_resident_fwd: a tensor of this call exceeds the 2 GB a buffer descriptor addresses");
    a.xproj = (const h16_t*)L->xproj; a.wh_t = (const h16_t*)L->wh_t; a.gates = (h16_t*)L->gates; a.c = L->c; a.h = (h16_t*)L->h; a.y = (h16_t*)L->y;
    a.mask = L->mask; a.hT = (h16_t*)L->hT; a.ld_hT = L->ld_hT; a.yT = (h16_t*)L->yT; a.ld_yT = L->ld_yT;
    a.T = T; a.B = B; a.kp = keep_prob;
    return MNN_OK;
}
// njobs layers of ONE shape and flavour (units 256, the same T, B, keep_prob, precision, mask / save choice) in one launch
extern "C" int mnn_lstm_resident_fwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_fwd_layer* L, float keep_prob) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && njobs >= 1 && njobs <= RES_MAX_JOBS, "mnn_lstm_resident_fwd_multi: 1..%d jobs", RES_MAX_JOBS);
    ResFwdJobs j{};
    j.njobs = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (int rc = res_fwd_fill(L + i, T, B, keep_prob, j.job[i])) return rc;
        MNN_REQUIRE((L[i].f16 != 0) == (L[0].f16 != 0) && (L[i].mask != nullptr) == (L[0].mask != nullptr) && (L[i].gates != nullptr) == (L[0].gates != nullptr),
                    "mnn_lstm_resident_fwd_multi: the jobs must share precision, dropout and save mode");
    }
    MNN_HIP(res_prepare());
    hipLaunchKernelGGL(res_fwd_kernel(L->f16 != 0, L->mask != nullptr, L->gates != nullptr), dim3(njobs * (B / 4)), dim3(256), ResGeom<256>::LDS, st, j);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" int mnn_lstm_resident_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob) {
    return mnn_lstm_resident_fwd_multi(s, T, B, 1, L, keep_prob);
}

static int res_bwd_fill(const mnn_lstm_bwd_layer* L, int T, int B, float keep_prob, ResBwdArgs& a) {
    MNN_REQUIRE(L && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_resident_bwd: bad arguments");
    MNN_REQUIRE(res_shape_ok(B, L->units), "mnn_lstm_resident_bwd: units must be 256 and B a multiple of 4 (B=%d u=%d)", B, L->units);
    MNN_REQUIRE(L->dh_ext && L->wh_p && L->gates && L->c, "mnn_lstm_resident_bwd: null pointer");
    MNN_REQUIRE(L->c0 == nullptr && L->dz == nullptr, "mnn_lstm_resident_bwd: no initial state / f32 dz output in this form");
    MNN_REQUIRE(L->dzT_t == nullptr || (L->ld_t == 0 ? (B & 31) == 0 : (L->ld_t >= T * B && (L->ld_t & 3) == 0)),
                "mnn_lstm_resident_bwd: ld_t too small / not a multiple of 4 (0 = the K-blocked layout [T*B/32][4u][32], B a multiple of 32)");
    MNN_REQUIRE((size_t)T * B * 256 * 8 < ((size_t)1 << 31) && (size_t)1024 * (size_t)L->ld_t * 2 < ((size_t)1 << 31),
                "mnn_lstm_resident_bwd: a tensor of this call exceeds the 2 GB a buffer descriptor addresses");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f), "mnn_lstm_resident_bwd: a keep mask goes with keep_prob < 1 and only with it (the forward's rule)");
    a.dh_ext = L->dh_ext; a.wh_p = (const h16_t*)L->wh_p; a.gates = (const h16_t*)L->gates; a.c = L->c; a.mask = keep_prob < 1.0f ? L->mask : nullptr;
    a.dzc = (h16_t*)L->dz_T; a.dzT = (h16_t*)L->dzT_t; a.ld_t = L->ld_t; a.db_p = L->db_p;
    a.T = T; a.B = B; a.kp = keep_prob;
    return MNN_OK;
}
extern "C" int mnn_lstm_resident_bwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_bwd_layer* L, float keep_prob) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && njobs >= 1 && njobs <= RES_MAX_JOBS, "mnn_lstm_resident_bwd_multi: 1..%d jobs", RES_MAX_JOBS);
    ResBwdJobs j{};
    j.njobs = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (int rc = res_bwd_fill(L + i, T, B, keep_prob, j.job[i])) return rc;
        MNN_REQUIRE((L[i].f16 != 0) == (L[0].f16 != 0) && (j.job[i].mask != nullptr) == (j.job[0].mask != nullptr),
                    "mnn_lstm_resident_bwd_multi: the jobs must share precision and dropout mode");
    }
    MNN_HIP(res_prepare());
    hipLaunchKernelGGL(res_bwd_kernel(L->f16 != 0, j.job[0].mask != nullptr), dim3(njobs * (B / 4)), dim3(256), ResBwdGeom<256>::LDS, st, j);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" int mnn_lstm_resident_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob) {
    return mnn_lstm_resident_bwd_multi(s, T, B, 1, L, keep_prob);
}
