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

#define RES_LDS_FENCE() asm volatile("" ::: "memory")
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
    static constexpr int XL = 0;                // + the fragments of tiles 0..3 at k-step KR - 1 (what is left of the 160 KB): 16 registers less
    static constexpr int WPW = NT * KL + XL;    // LDS fragments (1 KB each) per wave
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
#define RES_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void res_wait_all_but(int n) {       // loads, stores and LDS-DMA of a wave complete in issue order (MI355X_MICROARCH.md)
    switch (n) {
        RES_VMC(0) RES_VMC(1) RES_VMC(2) RES_VMC(3) RES_VMC(4) RES_VMC(5) RES_VMC(6) RES_VMC(7) RES_VMC(8) RES_VMC(9) RES_VMC(10) RES_VMC(11) RES_VMC(12)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int U, typename F, bool DROP, bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_res_fwd_kernel(ResFwdArgs A) {
    typedef ResGeom<U> G;
    typedef typename F::x8 frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 3, bank = (lane & 15) >> 2, g4 = lane >> 4;
    const int T = A.T, B = A.B;
    const int row0 = 4 * res_row_group(blockIdx.x, gridDim.x);
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
                if (s == G::KR - 1 && tl < G::XL) {
                    wl[(w * G::WPW + G::NT * G::KL + tl) * 64 + lane] = v;
                    wr[tl][s] = __builtin_bit_cast(frag_t, make_uint4(0u, 0u, 0u, 0u));          // never read
                } else if (s < G::KR) {
                    wr[tl][s] = __builtin_bit_cast(frag_t, v);
                    // k-steps 0..3 are pinned to AGPRs, which the matrix cores read directly (left to itself the register allocator treats
                    // AGPRs as spill space and copies every fragment back with four v_accvgpr_read per MFMA); 4..5 stay in VGPRs
                    if (s < 4) asm volatile("" : "+a"(wr[tl][s]));
                } else wl[(w * G::WPW + tl * G::KL + (s - G::KR)) * 64 + lane] = v;
            }
        }
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
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"((unsigned)(uintptr_t)(res_lptr_t)const_cast<void*>(l)), "v"(g) : "memory");
    };
    auto dma4 = [&](const void* g, const void* l) {
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"((unsigned)(uintptr_t)(res_lptr_t)const_cast<void*>(l)), "v"(g) : "memory");
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

    // ---- what leaves a step through LDS: h / y row-major (16-byte pieces) and h^T / y^T (a unit's four rows = 8 bytes) ----
    auto emit = [&](int t) {                       // outputs of step t: h[t] sits in state buffer (t + 1) & 1, y[t] in y buffer t & 1
        const char* hb = smem + G::OFF_H + ((t + 1) & 1) * 4 * G::PH;
        const char* yb = smem + G::OFF_Y + (t & 1) * 4 * G::PH;
        const int r = (tid >> 5) & 3, pc = tid & 31;
        if (DROP) {
            if (tid >= 128) *reinterpret_cast<uint4*>(A.y + (size_t)t * us + (size_t)(row0 + r) * U + pc * 8) = *reinterpret_cast<const uint4*>(yb + r * G::PH + pc * 16);
            else if (t + 1 == T) *reinterpret_cast<uint4*>(A.h + (size_t)t * us + (size_t)(row0 + r) * U + pc * 8) = *reinterpret_cast<const uint4*>(hb + r * G::PH + pc * 16);
        } else if (tid < 128) {
            *reinterpret_cast<uint4*>(A.h + (size_t)t * us + (size_t)(row0 + r) * U + pc * 8) = *reinterpret_cast<const uint4*>(hb + r * G::PH + pc * 16);
        }
        auto column = [&](const char* tile) {
            uint2 v;
            const h16_t* p = reinterpret_cast<const h16_t*>(tile) + tid;
            v.x = (unsigned)p[0] | ((unsigned)p[G::PH / 2] << 16);
            v.y = (unsigned)p[G::PH] | ((unsigned)p[3 * (G::PH / 2)] << 16);
            return v;
        };
        if (A.hT != nullptr && t + 1 < T) *reinterpret_cast<uint2*>(A.hT + (size_t)tid * A.ld_hT + (size_t)(t + 1) * B + row0) = column(hb);
        if (A.yT != nullptr) *reinterpret_cast<uint2*>(A.yT + (size_t)tid * A.ld_yT + (size_t)t * B + row0) = column(DROP ? yb : hb);
    };

    float creg[G::NG];
#pragma unroll
    for (int q = 0; q < G::NG; ++q) creg[q] = 0.f;
    const uint4* wl = reinterpret_cast<const uint4*>(smem + G::OFF_W) + (size_t)w * G::WPW * 64 + lane;
    // order of the k-steps inside a group: the two LDS-resident ones sit two register k-steps apart, so that the four fragment registers
    // of the first can be re-requested for the second while 8 MFMAs run
    constexpr int ORD[8] = {0, 1, 6, 2, 3, 7, 4, 5};
    constexpr int PFB = 3;                          // state fragments requested ahead of their MFMAs (LDS latency ~ 2 k-steps of 4 MFMAs)

    RES_TR_DECL;
    for (int t = 0; t < T; ++t) {
        RES_TR(0);
#ifdef RES_TRACE
        tr_[9] = clock64();
#endif
        if (t > 0) emit(t - 1);                                 // the previous step's outputs leave in the shadow of this step's MFMAs
        asm volatile("" ::: "memory");
        stage(t + 1 < T ? t + 1 : t, (t + 1) & 1);              // unconditional (clamped)
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
            auto in_lds = [&](int s_) { return s_ >= G::KR || (s_ == G::KR - 1 && 4 * q < G::XL); };
            auto lds_frag = [&](int s_, int j) { return s_ >= G::KR ? ((4 * q + j) * G::KL + (s_ - G::KR)) * 64 : (G::NT * G::KL + 4 * q + j) * 64; };
            frag_t b[G::KS], lw[4];
            int nxt = 0;                                     // position (in ORD) of the next LDS-resident k-step to request
            auto request = [&]() {
                while (nxt < G::KS && !in_lds(ORD[nxt])) ++nxt;
                if (nxt < G::KS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) lw[j] = __builtin_bit_cast(frag_t, wl[lds_frag(ORD[nxt], j)]);
                    ++nxt;
                }
            };
            request();
#pragma unroll
            for (int i = 0; i < PFB; ++i) b[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 64 * ORD[i]));
#pragma unroll
            for (int i = 0; i < G::KS; ++i) {
                const int s = ORD[i];
                if (i + PFB < G::KS) b[i + PFB] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 64 * ORD[i + PFB]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const mnn_f32x4 c0 = i == 0 ? mnn_f32x4{0.f, 0.f, 0.f, 0.f} : ac[j];
                    ac[j] = F::mfma16(in_lds(s) ? lw[j] : wr[4 * q + j][s < G::KR ? s : 0], b[i], c0);
                }
                if (in_lds(s)) request();
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
        res_wait_all_but(G::NG * (SAVE ? 2 : 1));               // the staged rows of step t + 1 are in LDS (the pointwise stores behind them may not be out)
        RES_TR(7);
        RES_BARRIER();
        RES_TR(8);
        RES_TR_FLUSH(0, t);
    }
    emit(T - 1);
}

extern "C" int mnn_lstm_resident_ok(int B, int units) {
    return (units == 256 && B > 0 && (B & 3) == 0) ? 1 : 0;
}

typedef void (*res_fwd_fn)(ResFwdArgs);
template <typename F> static res_fwd_fn res_fwd_pick(bool drop, bool save) {
    if (drop) return save ? lstm_res_fwd_kernel<256, F, true, true> : lstm_res_fwd_kernel<256, F, true, false>;
    return save ? lstm_res_fwd_kernel<256, F, false, true> : lstm_res_fwd_kernel<256, F, false, false>;
}
static res_fwd_fn res_fwd_kernel(bool f16, bool drop, bool save) { return f16 ? res_fwd_pick<Fp16F>(drop, save) : res_fwd_pick<Bf16F>(drop, save); }
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
    done[dev] = true;
    return hipSuccess;
}

extern "C" int mnn_lstm_resident_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_resident_fwd: bad arguments");
    MNN_REQUIRE(mnn_lstm_resident_ok(B, L->units), "mnn_lstm_resident_fwd: units must be 256 and B a multiple of 4 (B=%d u=%d)", B, L->units);
    MNN_REQUIRE(L->xproj && L->wh_t && L->c && L->h, "mnn_lstm_resident_fwd: null pointer");
    MNN_REQUIRE(L->xproj_bf16 != 0, "mnn_lstm_resident_fwd: the input projection must be in the layer's 16-bit type (gate-minor, bias included)");
    MNN_REQUIRE(L->h0 == nullptr && L->c0 == nullptr, "mnn_lstm_resident_fwd: an initial state is not supported by this form (zero state per window)");
    MNN_REQUIRE(L->hT == nullptr || (L->ld_hT >= T * B && (L->ld_hT & 3) == 0), "mnn_lstm_resident_fwd: ld_hT too small / not a multiple of 4");
    MNN_REQUIRE(L->yT == nullptr || (L->ld_yT >= T * B && (L->ld_yT & 3) == 0), "mnn_lstm_resident_fwd: ld_yT too small / not a multiple of 4");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                "mnn_lstm_resident_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    ResFwdArgs a{};
    a.xproj = (const h16_t*)L->xproj; a.wh_t = (const h16_t*)L->wh_t; a.gates = (h16_t*)L->gates; a.c = L->c; a.h = (h16_t*)L->h; a.y = (h16_t*)L->y;
    a.mask = L->mask; a.hT = (h16_t*)L->hT; a.ld_hT = L->ld_hT; a.yT = (h16_t*)L->yT; a.ld_yT = L->ld_yT;
    a.T = T; a.B = B; a.kp = keep_prob;
    MNN_HIP(res_prepare());
    hipLaunchKernelGGL(res_fwd_kernel(L->f16 != 0, L->mask != nullptr, L->gates != nullptr), dim3(B / 4), dim3(256), ResGeom<256>::LDS, st, a);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
