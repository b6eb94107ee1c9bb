// Cluster form of the CU-resident LSTM recurrence for a 512-unit layer (gfx950).  The layer's recurrent matrix is 2 MB in 16 bits: four times what
// one CU holds (lstm_resident.hip keeps a 256-unit layer's 512 KB on ONE CU).  Here EIGHT workgroups -- one per CU, all on one XCD -- share a tile
// of 32 batch rows for the whole sequence: member c keeps the weights of units [64 c, 64 c + 64) (256 gate columns x K = 512: 256 KB = 256
// registers per lane at one wave per SIMD, ALL of them AGPRs the matrix cores read in place; no weights in LDS), computes those units' gates and
// state for the 32 rows every step, and the members exchange h[t] (backward: partial sums of dh, see there) through a two-deep area that stays in
// their XCD's L2.  Forward:
//     wave: pointwise -> its 32 x 16 block of h[t] (1 KB) -> exchange area -> s_waitcnt -> its progress flag;
//           poll the cluster's 32 flags -> pull the whole 32 x 512 tile into LDS (LDS-DMA, L1-bypassing) -> workgroup barrier -> MFMAs.
// Against the row-parallel form (lstm_rowpar.hip: a workgroup = a 32-unit tile for every row tile, weights in LDS) a CU pulls 32 KB per step
// instead of 64 KB (backward: 32 KB out + 32 KB in instead of 256 KB in), reads no weight fragments from LDS, and every MFMA column carries a
// distinct row.  Measured at [1024, 256]: 4.3 / 4.8 us per timestep against 5.9 / 6.9 (profiles/round5_g_cluster_notes.md).
//
// Matrix-core layout (v_mfma_f32_32x32x16_{f16,bf16}; D[m][n] += A[m][k] B[k][n]; lane l holds A[l & 31][8 (l >> 5) ..+7], B[8 (l >> 5) ..+7][l & 31],
// D[8 (i >> 2) + 4 (l >> 5) + (i & 3)][l & 31], i = 0..15):
//   forward: A = weights, row m = 4 * (unit in tile) + gate (a tile = 8 units x {i, g, f, o}), B = h[t-1] (column n = batch row), so a lane's
//   accumulator quads are the four gates of the (row, unit) pairs  row = l & 31, unit = 8 j + 2 q + (l >> 5), q = 0..3: the gate pointwise runs
//   in the accumulator registers.  A wave owns two tiles (16 units), 32 k-steps each: 64 MFMAs per step (2048 cycles), 8 pairs per lane.
#include "common.h"
#include <stdlib.h>

typedef __attribute__((address_space(1))) unsigned cl_gu32;
typedef __attribute__((address_space(3))) void* cl_lptr_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#define CL_LIMIT 100000000LL       // spin bound: 1 s of wall_clock64() (100 MHz)
#define CL_FLAGS_OFF 32            // words: [0] status, [1] sticky, [32 + 32 cluster + 4 member + wave] progress flags (the row-parallel workspace layout)
#define CL_SC1 16                  // aux bit of raw buffer stores: device scope (write-through)
// aux bits of the streamed accesses: the row-major OUTPUTS (written once, read by a later kernel) are stored non-temporal so that they do not push
// the exchange lines out of the L2 (backward 5.0 -> 4.8 us per timestep); non-temporal LOADS of the operands were measured too and lose (6.2 us)
#define CL_NTL 0
#define CL_NTS 2
#define CL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define CL_FENCE() asm volatile("" ::: "memory")

#ifdef CL_TRACE     // development only (profiles/tools/cluster_trace.py): wall-clock stamps (100 MHz) of wave 0 of workgroup 0, [direction][step][stage]
__device__ long long cl_trace[2][512][12];
#define CL_TR(k) do { tr_[k] = wall_clock64(); } while (0)
#define CL_TR_DECL long long tr_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define CL_TR_FLUSH(dir, step) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (step) < 512) { _Pragma("unroll") for (int k_ = 0; k_ < 12; ++k_) cl_trace[dir][step][k_] = tr_[k_]; } } while (0)
extern "C" int mnn_lstm_cluster_trace(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cl_trace), sizeof(cl_trace)) == hipSuccess ? 0 : -2;
}
#else
#define CL_TR(k) do { } while (0)
#define CL_TR_DECL do { } while (0)
#define CL_TR_FLUSH(dir, step) do { } while (0)
#endif

__device__ __forceinline__ unsigned cl_ld(const unsigned* p) { return __hip_atomic_load((cl_gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cl_st(unsigned* p, unsigned v) { __hip_atomic_store((cl_gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// One wave waits until the first n (<= 32) words of `line` are all >= need; the upper lanes watch the status word.  false: the launch is aborting.
__device__ __forceinline__ bool cl_wait(const unsigned* line, unsigned* status, int n, unsigned need) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < n;
    const unsigned* p = mine ? line + lane : status;
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        const unsigned v = cl_ld(p);
        if (__all(mine ? v >= need : v == 0u)) return true;
        const bool dead = __any(!mine && v != 0u) || ((spins & 127u) == 0u && wall_clock64() - t0 > CL_LIMIT);
        if (dead) {
            if (lane == 0) { cl_st(status, 1u); cl_st(status + 1, 1u); }       // [1]: sticky, never re-zeroed by a launch
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// Launch start: do the eight workgroups of this cluster share an XCD?  Each posts 0x100 | XCC_ID (device scope), wave 0 waits for the others
// (bounded) and compares.  The answer selects the hand-off policy (results never depend on it): on one XCD the tiles and flags are stored
// write-BACK into a TWO-deep area -- they stay in that XCD's L2, which the consumers' L1-bypassing loads hit --, otherwise write-through into
// one slot per timestep (a line of another XCD's L2 is never re-used within the launch).
__device__ __forceinline__ void cl_probe_xcd(unsigned* xline, unsigned* status, int member, int* s_local) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) cl_st(xline + member, 0x100u | (xcc & 0xfu));
    if (threadIdx.x < 64) {
        bool same = false;
        if (cl_wait(xline, status, 8, 1u)) {
            const unsigned v = cl_ld(xline + (threadIdx.x < 8u ? threadIdx.x : 0));
            same = __all(v == __builtin_amdgcn_readfirstlane(v));
        }
        if (threadIdx.x == 0) *s_local = same ? 1 : 0;
    }
}
__device__ __forceinline__ void cl_raise(unsigned* flag, unsigned value, bool local) {        // one lane, after the wave's vmcnt(0)
    if (local) __hip_atomic_store((cl_gu32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store: lands in the shared L2
    else cl_st(flag, value);
}
#define CL_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void cl_wait_all_but(int n) {       // loads, stores and LDS-DMA of a wave complete in issue order (MI355X_MICROARCH.md)
    switch (n) {
        CL_VMC(0) CL_VMC(1) CL_VMC(2) CL_VMC(3) CL_VMC(4) CL_VMC(5) CL_VMC(6) CL_VMC(7) CL_VMC(8)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
// LDS-DMA in inline assembly (the builtin makes the compiler drain vmcnt(0) in front of every LDS read that may alias its destination:
// lstm_resident.hip); the sc1 form bypasses the L1 (the exchange area is re-written every other step)
__device__ __forceinline__ void cl_dma16(const void* g, const void* l) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"((unsigned)(uintptr_t)(cl_lptr_t)const_cast<void*>(l)), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void cl_dma16_sc1(const void* g, const void* l) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off sc1" :: "s"((unsigned)(uintptr_t)(cl_lptr_t)const_cast<void*>(l)), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void cl_dma4(const void* g, const void* l) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"((unsigned)(uintptr_t)(cl_lptr_t)const_cast<void*>(l)), "v"(g) : "memory", "m0");
}

struct ClFwdArgs {
    const h16_t* xproj; const h16_t* wh_t; h16_t* gates; float* c; h16_t* h; h16_t* y; const uint8_t* mask;
    h16_t* hT; int ld_hT; h16_t* yT; int ld_yT;
    char* xchg; unsigned* sync;
    int T, B, ncl, allow_local; float kp;
};
// Several independent layers of the same shape in ONE launch (mnn_lstm_cluster_fwd_multi / _bwd_multi: the per-track generators of the jamming
// mode, multinn_jamming.py:40-68).  A cluster -- eight workgroups, 32 rows of ONE job -- exchanges nothing with another cluster, so the jobs'
// clusters share the grid: gridDim.x = 8 * njobs * ncl; the grid's cluster c belongs to job c / ncl (its flags, exchange area and status word
// are that job's own workspace).  More clusters than the device has room for run in rounds: workgroups are dispatched in blockIdx order and a
// cluster's eight members are neighbours in that order, so a later round's cluster becomes resident as a whole when an earlier round drains.
#define CL_MAX_JOBS 8
struct ClFwdJobs { ClFwdArgs job[CL_MAX_JOBS]; int njobs; };

struct ClGeom {
    static constexpr int U = 512;
    static constexpr int PH = U * 2 + 16;            // pitch of a state row in LDS: rows 4 banks apart (the 16 rows of a ds_read_b128 phase cover all 64)
    static constexpr int HB = 32 * PH;               // one state buffer
    static constexpr int PX = 2 * 512 + 16;          // pitch of a PAIR of staged xproj rows (64 units x 8 bytes each; one LDS-DMA instruction writes a pair)
    static constexpr int XB = 16 * PX;
    static constexpr int MB = 32 * 64;               // keep bytes of the member's 64 units, 32 rows
    // wave-private output tiles (written element-wise in the accumulator layout, read back 16 bytes at a time in store order)
    static constexpr int PG = 16 * 8 + 16, PC = 16 * 4 + 16, PR = 16 * 2 + 16, PT = 32 * 2 + 16;
    static constexpr int S_G = 0, S_C = S_G + 32 * PG, S_H = S_C + 32 * PC, S_Y = S_H + 32 * PR, S_HT = S_Y + 32 * PR, S_YT = S_HT + 16 * PT, S_END = S_YT + 16 * PT;
    static constexpr int OFF_H = 0;
    static constexpr int OFF_X = OFF_H + 2 * HB;
    static constexpr int OFF_M = OFF_X + 2 * XB;
    static constexpr int OFF_S = OFF_M + 2 * MB;
    static constexpr int OFF_L = OFF_S + 4 * S_END;
    static constexpr int LDS = OFF_L + 16;
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <typename F, bool DROP, bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_cl_fwd_kernel(ClFwdJobs J) {
    typedef ClGeom G;
    typedef typename F::x8 frag_t;
    constexpr int U = G::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hf = lane >> 5;
    // workgroups are dealt round-robin over the 8 XCDs: XCD x = blockIdx & 7 takes the clusters (row tiles) [x ncl/8, (x+1) ncl/8) of the GRID, eight
    // consecutive workgroups of ITS sequence form a cluster; the grid's cluster then maps to (job, cluster of the job)
    const int nclg = J.job[0].ncl * J.njobs;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, mem = seq & 7, clg = xcd * (nclg >> 3) + (seq >> 3);
    const int jid = clg / J.job[0].ncl, cl = clg - jid * J.job[0].ncl;
    const ClFwdArgs A = J.job[jid];
    const int T = A.T, B = A.B;
    const int row0 = 32 * cl;
    const int ub = 64 * mem + 16 * w;                                           // first unit of this wave
    const size_t us = (size_t)B * U;
    const float ikp = 1.0f / A.kp;
    unsigned* status = A.sync;
    unsigned* flags = A.sync + CL_FLAGS_OFF + 32 * cl;
    int* s_local = reinterpret_cast<int*>(smem + G::OFF_L);
    cl_probe_xcd(A.sync + CL_FLAGS_OFF + 32 * A.ncl + 32 * cl, status, mem, s_local);

    // ---- the recurrent weights of this wave's 16 units: two tiles x 32 k-steps of A fragments, all pinned to AGPRs ----
    frag_t wr[2][32];
    {
        const int m = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int unit = ub + 8 * j + (m >> 2);
            const h16_t* src = A.wh_t + (size_t)gate_perm_col(m & 3, unit) * U + 8 * hf;
#pragma unroll
            for (int s = 0; s < 32; ++s) wr[j][s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(src + 16 * s));
        }
        // pinned only after ALL loads have been issued (a pin right behind its load makes every load wait for the one before: 64 round trips per launch)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 32; ++s) asm volatile("" : "+a"(wr[j][s]));
    }
    for (int i = tid; i < G::HB / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::OFF_H)[i] = 0u;       // h[-1] = 0 (state buffer 0)

    // ---- staging of a step's xproj rows (32 x 512 bytes) and keep bytes (32 x 64) a step ahead: wave w moves row pairs 4w..4w+3 / rows 8w..8w+7 ----
    const char* xsrc = reinterpret_cast<const char*>(A.xproj) + ((size_t)(row0 + hf) * U + 64 * mem) * 8 + (lane & 31) * 16;
    const uint8_t* msrc = DROP ? A.mask + (size_t)(row0 + (lane >> 4)) * U + 64 * mem + 4 * (lane & 15) : nullptr;
    auto stage = [&](int t, int buf) {
        const char* xb = xsrc + (size_t)t * us * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) cl_dma16(xb + (size_t)(2 * (4 * w + i)) * U * 8, smem + G::OFF_X + buf * G::XB + (4 * w + i) * G::PX);
        if (DROP) {
#pragma unroll
            for (int i = 0; i < 2; ++i) cl_dma4(msrc + (size_t)t * us + (size_t)(4 * (2 * w + i)) * U, smem + G::OFF_M + buf * G::MB + (2 * w + i) * 256);
        }
    };
    constexpr int NSTG = 4 + (DROP ? 2 : 0);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CL_BARRIER();
    const bool local = A.allow_local != 0 && *s_local != 0;

    // ---- the exchange area of this cluster: slots of 32 rows x 512 units (row-major, 1 KB per row) ----
    char* xch = A.xchg + (size_t)cl * (size_t)T * 32768;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xch, 0, (int)min((size_t)T * 32768, (size_t)0x7fffffff), 0x00020000);
    const unsigned vo_x = (unsigned)((lane >> 1) * 1024 + ub * 2 + (lane & 1) * 16);
    auto pull = [&](int slot, int buf) {                                        // rows 8w..8w+7 of the tile, one 1 KB row per instruction
        const char* src = xch + (size_t)slot * 32768 + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) cl_dma16_sc1(src + (8 * w + i) * 1024, smem + G::OFF_H + buf * G::HB + (8 * w + i) * G::PH);
    };

    // ---- outputs (buffer stores: scalar per-step base, one lane offset; a store whose lane offset is pushed out of range is dropped) ----
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)A.gates : (void*)A.c, 0, SAVE ? (int)min((size_t)T * us * 8, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)A.c, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)A.h, 0, (int)min((size_t)T * us * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(DROP ? (void*)A.y : (void*)A.h, 0, (int)min((size_t)T * us * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_hT = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)A.hT : (void*)A.h, 0, SAVE ? (int)min((size_t)U * A.ld_hT * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_yT = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)A.yT : (void*)A.h, 0, SAVE ? (int)min((size_t)U * A.ld_yT * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const unsigned vo_g = (unsigned)(((lane >> 3) * U + ub) * 8 + (lane & 7) * 16);          // + 8 i rows
    const unsigned vo_c = (unsigned)(((lane >> 2) * U + ub) * 4 + (lane & 3) * 16);          // + 16 i rows
    const unsigned vo_r = (unsigned)(((lane >> 1) * U + ub) * 2 + (lane & 1) * 16);
    const unsigned vo_hT = (unsigned)(((ub + (lane >> 2)) * (size_t)A.ld_hT + 8 * (lane & 3)) * 2);
    const unsigned vo_yT = (unsigned)(((ub + (lane >> 2)) * (size_t)A.ld_yT + 8 * (lane & 3)) * 2);
    char* sw = smem + G::OFF_S + w * G::S_END;                                   // this wave's output tiles

    float creg[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) creg[p] = 0.f;
    constexpr int PFB = 4;                          // state fragments requested ahead of their MFMAs
    CL_TR_DECL;
    for (int t = 0; t < T; ++t) {
        CL_TR(0);
        if (t > 0) {
            if (!cl_wait(flags, status, 32, (unsigned)t)) return;                // every wave of the cluster has stored its block of h[t-1]
            CL_TR(1);
            pull(local ? (t - 1) & 1 : t - 1, t & 1);
        }
        stage(t + 1 < T ? t + 1 : t, (t + 1) & 1);                              // unconditional (clamped)
        cl_wait_all_but(NSTG);                                                  // the pulled rows are in LDS (the staging of step t + 1 may still fly)
        CL_BARRIER();
        CL_TR(2);
        const char* hin = smem + G::OFF_H + (t & 1) * G::HB + row * G::PH + hf * 16;
        const char* xin = smem + G::OFF_X + (t & 1) * G::XB + (row >> 1) * G::PX + (row & 1) * 512 + (16 * w + hf) * 8;
        mnn_f32x16 acc[2];
        // tile 0 requests the 32 state fragments (PFB k-steps ahead of their MFMAs) and KEEPS them: tile 1 runs from registers -- a pure MFMA
        // stream for tile 0's pointwise to fill (left to itself the compiler reads every fragment a second time during tile 0)
        frag_t bq[32];
        auto tile = [&](int j, mnn_f32x16& ac) {
            if (j == 0) {
#pragma unroll
                for (int i = 0; i < PFB; ++i) bq[i] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 32 * i));
            }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (j == 0 && s + PFB < 32) bq[s + PFB] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(hin + 32 * (s + PFB)));
                mnn_f32x16 c0 = ac;
                if (s == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) c0[i] = 0.f;
                }
                ac = F::mfma32(wr[j][s], bq[s], c0);
                // vector / scalar / transcendental work, stores and LDS writes of the other tile's pointwise may cross; LDS reads and MFMAs may not
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x40 | 0x200);
            }
        };
        u32x2_t xv[4];
        u32x4_t mk16 = {0u, 0u, 0u, 0u};
        if (DROP) mk16 = *reinterpret_cast<const u32x4_t*>(smem + G::OFF_M + (t & 1) * G::MB + row * 64 + 16 * w);
        auto pointwise_pre = [&](int j) {                  // the tile's staged xproj values, requested in front of the other tile's MFMA stream
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const u32x2_t*>(xin + (8 * j + 2 * q) * 8);
        };
        auto pointwise = [&](int j, const mnn_f32x16& ac) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = 4 * j + q;
                const int u = 8 * j + 2 * q + hf;                               // unit inside the wave's 16
                const float gi = fast_sigmoid(ac[4 * q + 0] + F::lo(xv[q][0])), gg = fast_tanh(ac[4 * q + 1] + F::hi(xv[q][0]));
                const float gf = fast_sigmoid(ac[4 * q + 2] + F::lo(xv[q][1])), go = fast_sigmoid(ac[4 * q + 3] + F::hi(xv[q][1]));
                const float cv = gg * gi + creg[p] * gf;
                const float hv = fast_tanh(cv) * go;
                creg[p] = cv;
                const h16_t hb = F::cvt(hv);
                *reinterpret_cast<h16_t*>(sw + G::S_H + row * G::PR + u * 2) = hb;
                h16_t yb = hb;
                if (DROP) {
                    const unsigned kb = (mk16[(8 * j + 2 * q) >> 2] >> (8 * ((2 * q) & 3))) >> (8 * hf) & 0xffu;
                    yb = F::cvt(F::f32(hb) * ikp * (float)kb);
                    *reinterpret_cast<h16_t*>(sw + G::S_Y + row * G::PR + u * 2) = yb;
                }
                if (SAVE) {
                    u32x2_t pk;
                    pk[0] = pack2<F>(gi, gg);
                    pk[1] = pack2<F>(gf, go);
                    *reinterpret_cast<u32x2_t*>(sw + G::S_G + row * G::PG + u * 8) = pk;
                    *reinterpret_cast<h16_t*>(sw + G::S_HT + u * G::PT + row * 2) = hb;
                    if (DROP) *reinterpret_cast<h16_t*>(sw + G::S_YT + u * G::PT + row * 2) = yb;
                }
                *reinterpret_cast<float*>(sw + G::S_C + row * G::PC + u * 4) = cv;
            }
        };
        tile(0, acc[0]);
        pointwise_pre(0);
        tile(1, acc[1]);
        pointwise(0, acc[0]);
        pointwise_pre(1);
        pointwise(1, acc[1]);
        CL_TR(3);
        CL_FENCE();
        // ---- hand-off: this wave's 32 x 16 block of h[t] (LDS serves a wave's instructions in order: the element-wise writes above are visible) ----
        const u32x4_t hx = *reinterpret_cast<const u32x4_t*>(sw + G::S_H + (lane >> 1) * G::PR + (lane & 1) * 16);
        const unsigned so_x = (unsigned)((local ? t & 1 : t) * 32768);
        if (local) __builtin_amdgcn_raw_buffer_store_b128(hx, rs_x, vo_x, so_x, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(hx, rs_x, vo_x, so_x, CL_SC1);
        CL_FENCE();
        // the row-major outputs are read back while the hand-off store is on its way ...
        u32x4_t og[4], oc[2], oy = hx, ohT, oyT;
        if (SAVE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) og[i] = *reinterpret_cast<const u32x4_t*>(sw + G::S_G + (8 * i + (lane >> 3)) * G::PG + (lane & 7) * 16);
            ohT = *reinterpret_cast<const u32x4_t*>(sw + G::S_HT + (lane >> 2) * G::PT + (lane & 3) * 16);
            oyT = DROP ? *reinterpret_cast<const u32x4_t*>(sw + G::S_YT + (lane >> 2) * G::PT + (lane & 3) * 16) : ohT;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) oc[i] = *reinterpret_cast<const u32x4_t*>(sw + G::S_C + (16 * i + (lane >> 2)) * G::PC + (lane & 3) * 16);
        if (DROP) oy = *reinterpret_cast<const u32x4_t*>(sw + G::S_Y + (lane >> 1) * G::PR + (lane & 1) * 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // ... the hand-off store is out (nothing younger has been issued)
        if (lane == 0) cl_raise(flags + 4 * mem + w, (unsigned)(t + 1), local);
        CL_FENCE();
        CL_TR(4);
        // ... and leave behind the flag.  (Measured and dropped: holding them in registers and issuing them at the START of the next step, behind its
        // pull and staging requests, so that no store stands between a flag poll / pulled row and its wait -- 4.3 -> 5.6 us per timestep: the
        // issue of ten scattered 16-byte stores then sits on the chain in front of the barrier, here it runs under the exchange.)
        const unsigned so_e = (unsigned)((size_t)t * us + (size_t)row0 * U);      // element offset of the step's 32 x u block
        if (SAVE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(og[i], rs_g, vo_g + (unsigned)(8 * i * U * 8), so_e * 8, CL_NTS);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(oc[i], rs_c, vo_c + (unsigned)(16 * i * U * 4), so_e * 4, CL_NTS);
        if (DROP) {
            __builtin_amdgcn_raw_buffer_store_b128(oy, rs_y, vo_r, so_e * 2, CL_NTS);
            __builtin_amdgcn_raw_buffer_store_b128(hx, rs_h, vo_r | (t + 1 == T ? 0u : OOB), so_e * 2, CL_NTS);       // only the final state reads h
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(hx, rs_h, vo_r, so_e * 2, CL_NTS);
        }
        if (SAVE) {
            const int tn = t + 1 < T ? t + 1 : 0;
            __builtin_amdgcn_raw_buffer_store_b128(ohT, rs_hT, vo_hT | (t + 1 < T ? 0u : OOB), (unsigned)(((size_t)tn * B + row0) * 2), CL_NTS);
            __builtin_amdgcn_raw_buffer_store_b128(oyT, rs_yT, vo_yT, (unsigned)(((size_t)t * B + row0) * 2), CL_NTS);
        }
        CL_TR(5);
        CL_TR_FLUSH(0, t);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward:  dh = dh_ext[t] (/ kp * keep when a mask is given) + dz[t+1] . Wh ;  gate backward -> dz[t], dc   (rnn.py:124 LSTMBlockCell autodiff)
// The contraction runs over the 2048 gate columns.  Exchanging dz[t+1] itself (the forward's scheme) would make every CU pull the whole
// 32 x 2048 tile, 128 KB per step, on the chain.  Instead the K dimension is split the way the columns are OWNED: member c multiplies ITS OWN
// 256 columns of dz[t+1] -- which its waves wrote into its LDS one step earlier: no exchange in front of the MFMAs -- into partial sums for
// ALL 512 units (A = rows of Wh: 32 units per tile, B = the member's dz columns, K = 256: 4 tiles x 16 k-steps per wave), and the members
// reduce-scatter the partials: a wave stores, for each destination wave, the 8 values per lane that wave's lane needs (16-bit, 1 KB per
// store instruction, contiguous), raises its flag, polls the cluster's flags, loads the eight sources' 1 KB blocks of its own units and
// sums them in f32.  32 KB out + 32 KB in per CU and step.  Lane layout on both sides (v_mfma_f32_32x32x16 accumulator): row = l & 31,
// units 8 a + 4 (l >> 5) + b of a 32-unit tile, so a consumer lane owns two groups of four CONSECUTIVE units of one row: its operands
// (saved gates 32 bytes, c[t-1] 16, dh_ext 16, keep bytes 4 per group) are contiguous loads issued a step ahead.
// ------------------------------------------------------------------------------------------------------------------
struct ClBwdArgs {
    const float* dh_ext; const h16_t* wh_p; const h16_t* gates; const float* c; const uint8_t* mask;
    h16_t* dzc; h16_t* dzT; int ld_t; float* db_p;
    char* xchg; unsigned* sync;
    int T, B, ncl, allow_local; float kp;
};
struct ClBwdJobs { ClBwdArgs job[CL_MAX_JOBS]; int njobs; };

struct ClBwdGeom {
    static constexpr int U = 512;
    static constexpr int PZ = 256 * 2 + 16;          // pitch of a row of the member's dz columns in LDS (rows 4 banks apart)
    static constexpr int ZB = 32 * PZ;
    static constexpr int OFF_Z = 0;
    static constexpr int OFF_L = OFF_Z + 2 * ZB;
    static constexpr int LDS = OFF_L + 16;
    static constexpr int XBUF = 8 * 4 * 8 * 1024;    // one exchange buffer of a cluster: [destination member][destination wave][source member][lane] x 16 bytes
};

template <typename F, bool DROP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_cl_bwd_kernel(ClBwdJobs J) {
    typedef ClBwdGeom G;
    typedef typename F::x8 frag_t;
    constexpr int U = G::U;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = lane & 31, hf = lane >> 5;
    const int nclg = J.job[0].ncl * J.njobs;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, mem = seq & 7, clg = xcd * (nclg >> 3) + (seq >> 3);
    const int jid = clg / J.job[0].ncl, cl = clg - jid * J.job[0].ncl;
    const ClBwdArgs A = J.job[jid];
    const int T = A.T, B = A.B;
    const int row0 = 32 * cl;
    const size_t us = (size_t)B * U;
    const float ikp = 1.0f / A.kp;
    unsigned* status = A.sync;
    unsigned* flags = A.sync + CL_FLAGS_OFF + 32 * cl;
    int* s_local = reinterpret_cast<int*>(smem + G::OFF_L);
    cl_probe_xcd(A.sync + CL_FLAGS_OFF + 32 * A.ncl + 32 * cl, status, mem, s_local);

    // ---- producer side: Wh[unit][the member's 256 gate columns] for the units [128 w, 128 w + 128): four tiles x 16 k-steps of A fragments, in AGPRs ----
    frag_t wr[4][16];
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const h16_t* src = A.wh_p + (size_t)(128 * w + 32 * i + (lane & 31)) * (4 * U) + 256 * mem + 8 * hf;
#pragma unroll
            for (int s = 0; s < 16; ++s) wr[i][s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(src + 16 * s));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 16; ++s) asm volatile("" : "+a"(wr[i][s]));
    }
    for (int i = tid; i < G::ZB / 4; i += 256) reinterpret_cast<unsigned*>(smem + G::OFF_Z)[i] = 0u;       // dz[T] = 0

    // ---- consumer side: this lane's two groups of four units, their operands a step ahead (buffer loads: scalar per-step base, one lane offset) ----
    const int u0 = 64 * mem + 16 * w + 4 * hf;                                  // group a: units u0 + 8 a .. + 3
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)A.gates, 0, (int)min((size_t)T * us * 8, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)A.c, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)A.dh_ext, 0, (int)min((size_t)T * us * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(DROP ? (void*)A.mask : (void*)A.c, 0, (int)min((size_t)T * us, (size_t)0x7fffffff), 0x00020000);
    const unsigned vo_e = (unsigned)(row * U + u0);                             // element offset inside a step's 32 x u block (group a: + 8 a)
    u32x4_t gq0[2], gq1[2], cq[2], dq[2];
    unsigned mq[2] = {0u, 0u};
    auto request = [&](int t) {                                                 // operands of step t (t >= 0)
        const unsigned so = (unsigned)((size_t)t * us + (size_t)row0 * U);
        const unsigned sp = __builtin_amdgcn_readfirstlane((unsigned)((size_t)(t > 0 ? t - 1 : 0) * us + (size_t)row0 * U));
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            gq0[a] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (vo_e + 8 * a) * 8, so * 8, CL_NTL);
            gq1[a] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (vo_e + 8 * a) * 8 + 16, so * 8, CL_NTL);
            cq[a] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, (vo_e + 8 * a) * 4, sp * 4, CL_NTL);          // c[t-1] (t = 0: read and ignored)
            dq[a] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (vo_e + 8 * a) * 4, so * 4, CL_NTL);
            if (DROP) mq[a] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, vo_e + 8 * a, so, CL_NTL);
        }
    };
    float cnext[2][4], dcreg[2][4], dbv[2][4][4];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const u32x4_t c_last = __builtin_amdgcn_raw_buffer_load_b128(rs_c, (vo_e + 8 * a) * 4, (unsigned)((size_t)(T - 1) * us + (size_t)row0 * U) * 4, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cnext[a][r] = __uint_as_float(c_last[r]);
            dcreg[a][r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) dbv[a][g][r] = 0.f;
        }
    }
    request(T - 1);
    CL_BARRIER();
    const bool local = A.allow_local != 0 && *s_local != 0;
    if (*s_local == 0) {
        // the two-deep exchange area is re-written every other step: only valid while the cluster shares one L2.  Give up loudly (sticky status word).
        // (mnn_lstm_cluster_bwd_ok asks the placement on the host before this form is chosen: reaching this line means the probe and the launch disagree)
        if (tid == 0) { cl_st(status, 1u); cl_st(status + 1, 1u); }
        return;
    }

    // ---- the exchange area of this cluster (two buffers) ----
    char* xch = A.xchg + (size_t)cl * (size_t)(2 * G::XBUF);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xch, 0, 2 * G::XBUF, 0x00020000);
    const unsigned vo_xl = (unsigned)(((mem * 4 + w) * 8) * 1024 + lane * 16);   // loads: + 1024 source member

    // ---- outputs of a step (the member's 32 x 256 slice of dz, from the LDS tile): row-major 16-byte pieces and the transposed copy ----
    constexpr unsigned OOB = 0x80000000u;
    const size_t N = (size_t)T * B;
    const bool kb = A.ld_t == 0;
    const __amdgpu_buffer_rsrc_t rs_zc = __builtin_amdgcn_make_buffer_rsrc(A.dzc ? (void*)A.dzc : (void*)A.c, 0, A.dzc ? (int)min(N * 4 * U * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_zt = __builtin_amdgcn_make_buffer_rsrc(A.dzT ? (void*)A.dzT : (void*)A.c, 0,
                                                                           A.dzT ? (int)min(kb ? N * 4 * U * 2 : (size_t)4 * U * A.ld_t * 2, (size_t)0x7fffffff) : 0, 0x00020000);
    // piece j of this thread: p = tid + 256 j;  row-major: row p >> 5, 16-byte piece p & 31;  transposed: column p >> 2, rows 8 (p & 3) ..+7
    u32x4_t e_row;
    unsigned e_col[8];
    auto emit_read = [&](int buf, int j) {
        const char* zb = smem + G::OFF_Z + buf * G::ZB;
        const int p = tid + 256 * j;
        e_row = *reinterpret_cast<const u32x4_t*>(zb + (p >> 5) * G::PZ + (p & 31) * 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) e_col[k] = *reinterpret_cast<const h16_t*>(zb + (8 * (p & 3) + k) * G::PZ + (p >> 2) * 2);
    };
    auto emit_store = [&](int tt, int j) {          // dz[tt] (tt >= T: nothing)
        const unsigned none = tt >= T ? OOB : 0u;
        const int tc = tt >= T ? 0 : tt;
        const int p = tid + 256 * j;
        const unsigned so_c = (unsigned)(((size_t)tc * B + row0) * 4 * U * 2);
        __builtin_amdgcn_raw_buffer_store_b128(e_row, rs_zc, (unsigned)(((p >> 5) * 4 * U + 256 * mem) * 2 + (p & 31) * 16) | none, so_c, CL_NTS);
        u32x4_t v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = e_col[2 * k] | (e_col[2 * k + 1] << 16);
        const int col = 256 * mem + (p >> 2);
        const unsigned vo_t = kb ? (unsigned)(col * 64 + (p & 3) * 16) : (unsigned)(((size_t)col * A.ld_t + 8 * (p & 3)) * 2);
        const unsigned so_t = kb ? (unsigned)(((size_t)tc * (B >> 5) + cl) * (4 * U) * 64) : (unsigned)(((size_t)tc * B + row0) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_zt, vo_t | none, so_t, CL_NTS);
    };

    CL_TR_DECL;
    for (int kk = 0; kk < T; ++kk) {
        const int t = T - 1 - kk;
        CL_TR(0);
        const char* zin = smem + G::OFF_Z + (kk & 1) * G::ZB + row * G::PZ + hf * 16;
        char* zout = smem + G::OFF_Z + ((kk + 1) & 1) * G::ZB + row * G::PZ;
        const unsigned xb = (unsigned)((kk & 1) * G::XBUF);
        // ---- partial products, tile by tile: tile 0 requests the 16 fragments of the member's dz columns and keeps them; a tile's partials
        // leave (two 1 KB stores, one per destination wave) while the next tile runs ----
        frag_t bq[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mnn_f32x16 ac;
#pragma unroll
            for (int r = 0; r < 16; ++r) ac[r] = 0.f;
            if (i == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) bq[s] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(zin + 32 * s));
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (i == 0 && s + 4 < 16) bq[s + 4] = __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(zin + 32 * (s + 4)));
                ac = F::mfma32(wr[i][s], bq[s], ac);
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400 | 0x40 | 0x200);
            }
            // destination of half h of this tile: member 2 w + (i >> 1), wave 2 (i & 1) + h; its lane l reads what this lane l stores
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32x4_t v;
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = pack2<F>(ac[8 * h + 2 * k], ac[8 * h + 2 * k + 1]);
                const unsigned vo = (unsigned)(((((2 * w + (i >> 1)) * 4 + 2 * (i & 1) + h) * 8 + mem) * 1024) + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_x, vo, xb, 0);      // plain: stays in the XCD's L2 (any other placement has been refused above)
            }
        }
        CL_TR(1);
        // ---- everything of the gate backward that does not need dh is worked out while the partial sums travel: group 0 while this wave's
        // stores are on their way (the flag behind them then goes up without a wait), group 1 behind the flag (under the poll) ----
        float pe[2][4], pA[2][4], pK[2][4][4], pf[2][4], pc[2][4];             // dh_ext share, d c / d h, d z_gate / d c (gate o: / d h), f, c[t-1]
        auto pre = [&](int a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned g01 = r < 2 ? gq0[a][2 * r] : gq1[a][2 * r - 4], g23 = r < 2 ? gq0[a][2 * r + 1] : gq1[a][2 * r - 3];
                const float gi = F::lo(g01), gg = F::hi(g01), gf = F::lo(g23), go = F::hi(g23);
                const float dv = __uint_as_float(dq[a][r]);
                pe[a][r] = DROP ? dv * ikp * (float)((mq[a] >> (8 * r)) & 0xffu) : dv;
                const float tc = fast_tanh(cnext[a][r]);
                const float cprev = t > 0 ? __uint_as_float(cq[a][r]) : 0.f;
                pA[a][r] = go * (1.f - tc * tc);
                pK[a][0][r] = gg * gi * (1.f - gi);
                pK[a][1][r] = gi * (1.f - gg * gg);
                pK[a][2][r] = cprev * gf * (1.f - gf);
                pK[a][3][r] = tc * go * (1.f - go);
                pf[a][r] = gf;
                pc[a][r] = cprev;
            }
        };
        __builtin_amdgcn_sched_barrier(0);
        pre(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // this wave's partials are out
        if (lane == 0) cl_raise(flags + 4 * mem + w, (unsigned)(kk + 1), local);
        CL_TR(2);
        __builtin_amdgcn_sched_barrier(0);
        pre(1);
        __builtin_amdgcn_sched_barrier(0);
        if (!cl_wait(flags, status, 32, (unsigned)(kk + 1))) return;             // every wave of the cluster has stored its partials of this step
        CL_TR(3);
        float dhr[2][4];
        u32x4_t pq[8];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) pq[s8] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo_xl + 1024 * s8, xb, CL_SC1);
        __builtin_amdgcn_sched_barrier(0);
        // the previous step's outputs (dz[t+1], still in the tile the MFMAs read) leave HERE, behind the requests for the partial sums and while
        // those travel: in the MFMA stream, in front of the partial stores, they delayed the flag (a wave's stores are acknowledged in issue
        // order) -- 5.33 -> 5.01 us per timestep at [1024, 256]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            emit_read(kk & 1, j);
            emit_store(t + 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float sum = 0.f;
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    const unsigned wv = pq[s8][2 * a + (r >> 1)];
                    sum += (r & 1) ? F::hi(wv) : F::lo(wv);
                }
                dhr[a][r] = sum;
            }
        CL_TR(4);
        // ---- what is left once dh is known: register r of group a = unit u0 + 8 a + r ----
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            h16_t b4[4][4];                                 // [gate][unit]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = pe[a][r] + dhr[a][r];
                const float d_c = dh * pA[a][r] + dcreg[a][r];
                const float dzv[4] = {d_c * pK[a][0][r], d_c * pK[a][1][r], d_c * pK[a][2][r], dh * pK[a][3][r]};
                dcreg[a][r] = d_c * pf[a][r];
                cnext[a][r] = pc[a][r];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    b4[g][r] = F::cvt(dzv[g]);
                    dbv[a][g][r] += F::f32(b4[g][r]);       // the (16-bit) values the weight-gradient GEMMs see
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2_t v;
                v[0] = (unsigned)b4[g][0] | ((unsigned)b4[g][1] << 16);
                v[1] = (unsigned)b4[g][2] | ((unsigned)b4[g][3] << 16);
                *reinterpret_cast<u32x2_t*>(zout + (gate_perm_col(g, u0 + 8 * a) - 256 * mem) * 2) = v;
            }
        }
        CL_TR(5);
        request(t > 0 ? t - 1 : 0);                     // unconditional (clamped): behind a condition the compiler COPIES the freshly requested registers at the join -- and waits for the loads to do it
        CL_BARRIER();
        CL_TR(6);
        CL_TR_FLUSH(1, kk);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        emit_read(T & 1, j);
        emit_store(0, j);
    }
    if (A.db_p != nullptr) {                            // bias gradient: sums over the cluster's 32 rows and all steps
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = dbv[a][g][r];
#pragma unroll
                    for (int d = 1; d < 32; d <<= 1) v += __shfl_xor(v, d);
                    if (row == 0) atomicAdd(A.db_p + gate_perm_col(g, u0 + 8 * a + r), v);
                }
    }
}

// ---------------------------------------------------------------------------------------------- host side
void mnn_rp_workspace_layout(int nrt, int U, size_t* sync_bytes, size_t* xchg_off);           // lstm_rowpar.hip: the workspace both forms share
int mnn_rp_reset_launch(hipStream_t st, void* workspace, int nrt, int U);

static int cl_cu_count() {
    static int n[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (n[dev] <= 0) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        n[dev] = p.multiProcessorCount;
    }
    return n[dev];
}
// 32 rows per cluster of eight workgroups, every workgroup on its own CU, the clusters dealt evenly over the 8 XCDs
static bool cl_shape_ok(int B, int units) { return units == 512 && B > 0 && (B % 256) == 0 && (B / 32) * 8 <= cl_cu_count(); }

typedef void (*cl_fwd_fn)(ClFwdJobs);
template <typename F> static cl_fwd_fn cl_fwd_pick(bool drop, bool save) {
    if (drop) return save ? lstm_cl_fwd_kernel<F, true, true> : lstm_cl_fwd_kernel<F, true, false>;
    return save ? lstm_cl_fwd_kernel<F, false, true> : lstm_cl_fwd_kernel<F, false, false>;
}
static cl_fwd_fn cl_fwd_kernel(bool f16, bool drop, bool save) { return f16 ? cl_fwd_pick<Fp16F>(drop, save) : cl_fwd_pick<Bf16F>(drop, save); }
typedef void (*cl_bwd_fn)(ClBwdJobs);
static cl_bwd_fn cl_bwd_kernel(bool f16, bool drop) {
    if (f16) return drop ? lstm_cl_bwd_kernel<Fp16F, true> : lstm_cl_bwd_kernel<Fp16F, false>;
    return drop ? lstm_cl_bwd_kernel<Bf16F, true> : lstm_cl_bwd_kernel<Bf16F, false>;
}
static hipError_t cl_prepare() {
    static bool done[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    for (int i = 0; i < 8; ++i) {
        e = hipFuncSetAttribute((const void*)cl_fwd_kernel(i & 1, i & 2, i & 4), hipFuncAttributeMaxDynamicSharedMemorySize, ClGeom::LDS);
        if (e != hipSuccess) return e;
    }
    for (int i = 0; i < 4; ++i) {
        e = hipFuncSetAttribute((const void*)cl_bwd_kernel(i & 1, i & 2), hipFuncAttributeMaxDynamicSharedMemorySize, ClBwdGeom::LDS);
        if (e != hipSuccess) return e;
    }
    done[dev] = true;
    return hipSuccess;
}
// (also raises the kernels' dynamic-LDS limit on this device: the host asks before every use, so never for the first time under stream capture)
extern "C" int mnn_lstm_cluster_ok(int B, int units) {
    if (!cl_shape_ok(B, units)) return 0;
    return cl_prepare() == hipSuccess ? 1 : 0;
}

// Placement probe: is every cluster of a (8 ncl)-workgroup launch of these kernels dealt onto ONE XCD?  The backward's two-deep exchange area is only
// valid then (the forward has a write-through fall-back, the backward does not), and the answer is a property of the device's partitioning and the
// dispatcher's dealing order -- deterministic, so it is asked ONCE per device and grid on the host (a launch of the kernels' own grid and LDS size whose
// workgroups post their XCC id), not found out by a training step that has already consumed unwritten gradients.
__global__ void __launch_bounds__(256) cl_place_probe_kernel(unsigned* __restrict__ out) {
    extern __shared__ char cl_probe_smem_[];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        cl_probe_smem_[0] = (char)xcc;                       // (the dynamic LDS is what keeps one workgroup per CU, as in the real launch)
        out[blockIdx.x] = 0x100u | (xcc & 0xfu);
    }
}
static int cl_placement_ok(int ncl) {
    static signed char known[64][9];                         // [device][ncl / 8]: 0 unknown, 1 every cluster on one XCD, -1 not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || ncl < 8 || ncl > 64 || (ncl & 7)) return 0;
    signed char& k = known[dev][ncl / 8];
    if (k != 0) return k > 0;
    // may be reached for the first time while ANOTHER stream of this thread is capturing: the probe runs on its own stream, relaxed mode for its calls
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) return 0;
    int ok = 0;
    hipStream_t st = nullptr;
    unsigned* d = nullptr;
    unsigned host[512];
    do {
        if (hipFuncSetAttribute((const void*)cl_place_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ClBwdGeom::LDS) != hipSuccess) break;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        if (hipMalloc((void**)&d, 8 * ncl * sizeof(unsigned)) != hipSuccess) break;
        if (hipMemsetAsync(d, 0, 8 * ncl * sizeof(unsigned), st) != hipSuccess) break;
        hipLaunchKernelGGL(cl_place_probe_kernel, dim3(8 * ncl), dim3(256), ClBwdGeom::LDS, st, d);
        if (hipGetLastError() != hipSuccess) break;
        if (hipMemcpyAsync(host, d, 8 * ncl * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (hipStreamSynchronize(st) != hipSuccess) break;
        ok = 1;
        for (int b = 0; b < 8 * ncl && ok > 0; ++b) {        // the kernels' own mapping: cluster = (blockIdx & 7) * (ncl / 8) + (blockIdx >> 6)
            const int xcd = b & 7, seq = b >> 3, cl = xcd * (ncl >> 3) + (seq >> 3);
            const int b0 = xcd + 8 * (8 * (cl - xcd * (ncl >> 3)));     // member 0 of the same cluster
            if ((host[b] & 0x100u) == 0u || host[b] != host[b0]) ok = -1;
        }
    } while (false);
    if (d) (void)hipFree(d);
    if (st) (void)hipStreamDestroy(st);
    (void)hipThreadExchangeStreamCaptureMode(&mode);
    if (ok == 0) return 0;                                   // the probe itself failed: unknown stays unknown, answer "no"
    k = (signed char)ok;
    return ok > 0;
}
// The backward may run: shape covered AND every cluster sits on one XCD (MNN_PERSIST_NO_LOCAL, the tests' hook, answers "no" like a repartitioned device).
// The caller falls back to mnn_lstm_rowpar_bwd, which reads the same saved activations.
extern "C" int mnn_lstm_cluster_bwd_ok(int B, int units) {
    if (!mnn_lstm_cluster_ok(B, units)) return 0;
    if (getenv("MNN_PERSIST_NO_LOCAL") != nullptr) return 0;
    return cl_placement_ok(B / 32);
}
// ... for a launch of njobs layers (mnn_lstm_cluster_bwd_multi: the grid's clusters, later rounds included, are probed as one grid)
extern "C" int mnn_lstm_cluster_bwd_multi_ok(int B, int units, int njobs) {
    if (!mnn_lstm_cluster_ok(B, units) || njobs < 1 || njobs > CL_MAX_JOBS || ((njobs * (B / 32)) & 7) != 0) return 0;
    if (getenv("MNN_PERSIST_NO_LOCAL") != nullptr) return 0;
    return cl_placement_ok(njobs * (B / 32));
}

static int cl_fwd_fill(const mnn_lstm_fwd_layer* L, int T, int B, float keep_prob, void* workspace, ClFwdArgs& a) {
    MNN_REQUIRE(L && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_cluster_fwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm_cluster_fwd: workspace must be 256-byte aligned");
    MNN_REQUIRE(cl_shape_ok(B, L->units), "mnn_lstm_cluster_fwd: units must be 512, B a multiple of 256 and B / 4 at most the device's CUs (B=%d u=%d)", B, L->units);
    MNN_REQUIRE(L->xproj && L->wh_t && L->c && L->h, "mnn_lstm_cluster_fwd: null pointer");
    MNN_REQUIRE(L->xproj_bf16 != 0, "mnn_lstm_cluster_fwd: the input projection must be in the layer's 16-bit type (gate-minor, bias included)");
    MNN_REQUIRE(L->h0 == nullptr && L->c0 == nullptr, "mnn_lstm_cluster_fwd: an initial state is not supported by this form (zero state per window)");
    MNN_REQUIRE(L->hT == nullptr || (L->ld_hT >= T * B && (L->ld_hT & 7) == 0), "mnn_lstm_cluster_fwd: ld_hT too small / not a multiple of 8");
    MNN_REQUIRE(L->yT == nullptr || (L->ld_yT >= T * B && (L->ld_yT & 7) == 0), "mnn_lstm_cluster_fwd: ld_yT too small / not a multiple of 8");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                "mnn_lstm_cluster_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    MNN_REQUIRE((L->gates != nullptr) == (L->hT != nullptr) && (L->gates != nullptr) == (L->yT != nullptr),
                "mnn_lstm_cluster_fwd: the saved gates, hT and yT come together (training) or not at all");
    MNN_REQUIRE((size_t)T * B * 512 * 8 < ((size_t)1 << 31) && (size_t)512 * (size_t)(L->ld_hT > L->ld_yT ? L->ld_hT : L->ld_yT) * 2 < ((size_t)1 << 31),
                "mnn_lstm_cluster_fwd: a tensor of this call exceeds the 2 GB a buffer descriptor addresses");
    a.xproj = (const h16_t*)L->xproj; a.wh_t = (const h16_t*)L->wh_t; a.gates = (h16_t*)L->gates; a.c = L->c; a.h = (h16_t*)L->h; a.y = (h16_t*)L->y;
    a.mask = L->mask; a.hT = (h16_t*)L->hT; a.ld_hT = L->ld_hT; a.yT = (h16_t*)L->yT; a.ld_yT = L->ld_yT;
    size_t sync_bytes = 0, xoff = 0;
    mnn_rp_workspace_layout(B / 32, 512, &sync_bytes, &xoff);
    a.sync = (unsigned*)workspace; a.xchg = (char*)workspace + xoff;
    a.T = T; a.B = B; a.ncl = B / 32; a.kp = keep_prob; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    return MNN_OK;
}
// njobs layers of ONE shape and flavour in one launch (see ClFwdJobs); workspaces: one lstm_rowpar workspace per job
extern "C" int mnn_lstm_cluster_fwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_fwd_layer* L, float keep_prob, void* const* workspaces) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && workspaces && njobs >= 1 && njobs <= CL_MAX_JOBS, "mnn_lstm_cluster_fwd_multi: 1..%d jobs", CL_MAX_JOBS);
    ClFwdJobs j{};
    j.njobs = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (int rc = cl_fwd_fill(L + i, T, B, keep_prob, workspaces[i], j.job[i])) return rc;
        MNN_REQUIRE((L[i].f16 != 0) == (L[0].f16 != 0) && (L[i].mask != nullptr) == (L[0].mask != nullptr) && (L[i].gates != nullptr) == (L[0].gates != nullptr),
                    "mnn_lstm_cluster_fwd_multi: the jobs must share precision, dropout and save mode");
        for (int k = 0; k < i; ++k) MNN_REQUIRE(workspaces[k] != workspaces[i], "mnn_lstm_cluster_fwd_multi: every job needs its own workspace");
    }
    MNN_REQUIRE(((njobs * (B / 32)) & 7) == 0, "mnn_lstm_cluster_fwd_multi: the grid's clusters must be a multiple of 8");
    MNN_HIP(cl_prepare());
    for (int i = 0; i < njobs; ++i)
        if (int rc = mnn_rp_reset_launch(st, workspaces[i], B / 32, 512)) return rc;
    hipLaunchKernelGGL(cl_fwd_kernel(L->f16 != 0, L->mask != nullptr, L->gates != nullptr), dim3(8 * njobs * (B / 32)), dim3(256), ClGeom::LDS, st, j);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" int mnn_lstm_cluster_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob, void* workspace) {
    return mnn_lstm_cluster_fwd_multi(s, T, B, 1, L, keep_prob, &workspace);
}

static int cl_bwd_fill(const mnn_lstm_bwd_layer* L, int T, int B, float keep_prob, void* workspace, ClBwdArgs& a) {
    MNN_REQUIRE(L && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_cluster_bwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm_cluster_bwd: workspace must be 256-byte aligned");
    MNN_REQUIRE(cl_shape_ok(B, L->units), "mnn_lstm_cluster_bwd: units must be 512, B a multiple of 256 and B / 4 at most the device's CUs (B=%d u=%d)", B, L->units);
    MNN_REQUIRE(L->dh_ext && L->wh_p && L->gates && L->c, "mnn_lstm_cluster_bwd: null pointer");
    MNN_REQUIRE(L->c0 == nullptr && L->dz == nullptr, "mnn_lstm_cluster_bwd: no initial state / f32 dz output in this form");
    MNN_REQUIRE(L->dzT_t == nullptr || L->ld_t == 0 || (L->ld_t >= T * B && (L->ld_t & 7) == 0),
                "mnn_lstm_cluster_bwd: ld_t too small / not a multiple of 8 (0 = the K-blocked layout [T*B/32][4u][32])");
    MNN_REQUIRE((size_t)T * B * 512 * 8 < ((size_t)1 << 31) && (size_t)2048 * (size_t)L->ld_t * 2 < ((size_t)1 << 31),
                "mnn_lstm_cluster_bwd: a tensor of this call exceeds the 2 GB a buffer descriptor addresses");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f), "mnn_lstm_cluster_bwd: a keep mask goes with keep_prob < 1 and only with it (the forward's rule)");
    MNN_REQUIRE(T >= 4, "mnn_lstm_cluster_bwd: T >= 4 (the two exchange buffers of a cluster take the room of four timesteps of the row-parallel workspace)");
    a.dh_ext = L->dh_ext; a.wh_p = (const h16_t*)L->wh_p; a.gates = (const h16_t*)L->gates; a.c = L->c; a.mask = keep_prob < 1.0f ? L->mask : nullptr;
    a.dzc = (h16_t*)L->dz_T; a.dzT = (h16_t*)L->dzT_t; a.ld_t = L->ld_t; a.db_p = L->db_p;
    size_t sync_bytes = 0, xoff = 0;
    mnn_rp_workspace_layout(B / 32, 512, &sync_bytes, &xoff);
    a.sync = (unsigned*)workspace; a.xchg = (char*)workspace + xoff;
    a.T = T; a.B = B; a.ncl = B / 32; a.kp = keep_prob; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    return MNN_OK;
}
extern "C" int mnn_lstm_cluster_bwd_multi(mnn_stream_t s, int T, int B, int njobs, const mnn_lstm_bwd_layer* L, float keep_prob, void* const* workspaces) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && workspaces && njobs >= 1 && njobs <= CL_MAX_JOBS, "mnn_lstm_cluster_bwd_multi: 1..%d jobs", CL_MAX_JOBS);
    ClBwdJobs j{};
    j.njobs = njobs;
    for (int i = 0; i < njobs; ++i) {
        if (int rc = cl_bwd_fill(L + i, T, B, keep_prob, workspaces[i], j.job[i])) return rc;
        MNN_REQUIRE((L[i].f16 != 0) == (L[0].f16 != 0) && (j.job[i].mask != nullptr) == (j.job[0].mask != nullptr),
                    "mnn_lstm_cluster_bwd_multi: the jobs must share precision and dropout mode");
        for (int k = 0; k < i; ++k) MNN_REQUIRE(workspaces[k] != workspaces[i], "mnn_lstm_cluster_bwd_multi: every job needs its own workspace");
    }
    MNN_REQUIRE(((njobs * (B / 32)) & 7) == 0, "mnn_lstm_cluster_bwd_multi: the grid's clusters must be a multiple of 8");
    MNN_HIP(cl_prepare());
    for (int i = 0; i < njobs; ++i)
        if (int rc = mnn_rp_reset_launch(st, workspaces[i], B / 32, 512)) return rc;
    hipLaunchKernelGGL(cl_bwd_kernel(L->f16 != 0, j.job[0].mask != nullptr), dim3(8 * njobs * (B / 32)), dim3(256), ClBwdGeom::LDS, st, j);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" int mnn_lstm_cluster_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob, void* workspace) {
    return mnn_lstm_cluster_bwd_multi(s, T, B, 1, L, keep_prob, &workspace);
}
