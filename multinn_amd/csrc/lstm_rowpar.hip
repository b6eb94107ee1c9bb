// Row-parallel persistent LSTM recurrence for gfx950 (bf16 operands, f32 state): ONE launch runs all T steps of ONE layer.
//
// The two-layer persistent form of lstm_persist.hip splits K over the waves of a workgroup (weights in registers) and pays, per 32-row
// item, a cross-wave reduction through LDS, three workgroup barriers and a poll: ~3.7 us whatever the item's height.  At B >= 512 a
// workgroup walks several row tiles per timestep and that fixed cost is paid once per tile (TGT [1024,256,88,5]: 4 tiles -> 15.6 us per
// timestep forward, 21.9 us backward, for 1.7 us of MFMA work).  Here the decomposition is turned round:
//   * a workgroup owns one 32-unit tile (128 gate columns) of the layer and keeps that slice of the recurrent weights in LDS, stored in
//     MFMA B-fragment order (one conflict-free 1-KiB ds_read_b128 per MFMA): 128 KiB for 512 units;
//   * a WAVE owns one 32-row tile for the whole sequence and is an independent agent: it polls the progress flags of its row tile, pulls
//     the previous step's state tile (A fragments, straight to registers), runs the full-K product for its 128 columns against the
//     weights in LDS, does the gate pointwise IN REGISTERS (the four gates of a (row, unit) sit in the same lane: gate-interleaved
//     columns), hands its h / dz tile on, and raises its own flag.  No K split, no cross-wave reduction, no workgroup barrier after
//     the weights are loaded;
//   * the layers run as separate launches (the input projection of the next layer is one large GEMM between them), which keeps K at
//     the layer's own width: backward, 4U = 2048 columns of dz for 512 units fit LDS (128 KiB); the fused K = 4 U1 + 4 U2 does not.
// Hand-off protocol: the one of lstm_persist.hip (cdna_hip_programming.md G16 R1; MI355X_MICROARCH.md "Valid forms", first table row)
// with the workgroup replaced by the wave: 16-byte sc1 (write-through) stores of the tile in A-fragment order -> the storing wave's
// `s_waitcnt vmcnt(0)` -> ONE sc1 flag store by that wave; the consumer wave polls the flags of its row tile with sc1 loads and only
// then issues its sc1 loads of the tile.  Every spin is bounded and watches a status word; a give-up is sticky (mnn_lstm_rowpar_status).
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

#define RP_LIMIT 100000000LL       // spin bound: 1 s of wall_clock64() (100 MHz)
#define RP_FLAGS_OFF 32            // words: [0] status, [1] sticky, [32 + 32 rt + member] progress flags
#ifndef RP_STG
#define RP_STG 33   // f32 words per staged row (32 + pad)
#endif
#ifndef RP_RING
#define RP_RING 3                  // backward: chunks of the dz tile in the register ring (8 KiB each)
#endif
#define RP_SC1 16                  // aux bit of raw buffer loads/stores: device scope (write-through / L1-bypassing)
#ifndef RP_A_AUX
#define RP_A_AUX 0                  // pair forward, same-XCD mode: state-tile loads through the L1
#endif
// The per-wave LDS tiles are written element-wise (bf16) and read back 16 bytes at a time by OTHER lanes of the same wave: the LDS serves a
// wave's instructions in order, but the compiler must not move the differently-typed accesses across one another
#define RP_LDS_FENCE() asm volatile("" ::: "memory")
// Behind the hand-off stores: rp_wait_all_but(N) counts the vector-memory operations written AFTER them in the source, so neither the
// scheduler nor the memory-dependence analysis may move one of those above the hand-off (vmcnt retires in issue order)
// (the marker comment lets tests/test_build_guards.py find the fence in the compiler's output)
#define RP_HANDOFF_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("; RP_HANDOFF_FENCE" ::: "memory"); } while (0)

#ifdef RP_TRACE     // development only (profiles/tools/rowpar_trace.py): wall-clock stamps of one wave per kernel, [direction][step][stage]
__device__ long long rp_trace[2][512][12];
#define RP_TR(dir, cond, step, k) do { if ((cond) && (step) < 512 && (threadIdx.x & 63) == 0) rp_trace[dir][step][k] = wall_clock64(); } while (0)
extern "C" int mnn_lstm_rowpar_trace(void* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rp_trace), sizeof(rp_trace)) == hipSuccess ? 0 : -2;
}
#else
#define RP_TR(dir, cond, step, k) do { } while (0)
#endif

__device__ __forceinline__ unsigned rp_ld(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rp_st(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rp_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// fragment row of C register k in lane half hh (v_mfma_f32_32x32x16_bf16): (k & 3) + 8 (k >> 2) + 4 hh
__device__ __forceinline__ constexpr int rp_krow(int k) { return (k & 3) + 8 * (k >> 2); }

// One wave waits until the first n words of `line` are all >= need; the other lanes watch the status word.  false: the launch is aborting.
__device__ __forceinline__ bool rp_wait(const unsigned* line, unsigned* status, int n, unsigned need) {
    const int lane = threadIdx.x & 63;
    const bool mine = lane < n;
    const unsigned* p = mine ? line + lane : status;
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        const unsigned v = rp_ld(p);
        if (__all(mine ? v >= need : v == 0u)) return true;
        const bool dead = __any(!mine && v != 0u) || ((spins & 127u) == 0u && wall_clock64() - t0 > RP_LIMIT);
        if (dead) {
            if (lane == 0) { rp_st(status, 1u); rp_st(status + 1, 1u); }       // [1]: sticky, never re-zeroed by a launch
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

struct RFwdArgs {
    const float* xproj; const bf16_t* wh_t; bf16_t* gates; float* c; bf16_t* h; bf16_t* y; const uint8_t* mask;
    bf16_t* hT; int ld_hT; bf16_t* yT; int ld_yT;
    char* hx; const char* hx0; unsigned* sync;
    int T, B, nrt, G, allow_local; float kp;
};
// xproj as stored: f32 float4 (i, g, f, o) per (row, unit), or four bf16 in 8 bytes (template parameter XB of the forward kernels)
template <bool XB, typename F> struct RpX;
template <typename F> struct RpX<false, F> {
    typedef float4 T;
    static __device__ __forceinline__ float4 get(const float4& v) { return v; }
};
template <typename F> struct RpX<true, F> {
    typedef uint2 T;
    static __device__ __forceinline__ float4 get(const uint2& v) { return make_float4(F::lo(v.x), F::hi(v.x), F::lo(v.y), F::hi(v.y)); }
};

// Launch start: do the U/32 workgroups of this row-tile group share an XCD?  Each posts 0x100 | XCC_ID (device scope), wave 0 waits for the
// others (bounded) and compares; the answer reaches the other waves through LDS at the barrier that follows the weight load.  It only
// selects the store policy of the hand-offs (results never depend on it): on one XCD the tile and the flag are stored write-BACK -- they stay
// in that XCD's L2, which the consumers' sc1 (L1-bypassing, L2-served) loads hit -- otherwise write-through (device scope).
__device__ __forceinline__ void rp_probe_xcd(unsigned* xline, unsigned* status, int member, int nb, int* s_local) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) rp_st(xline + member, 0x100u | (xcc & 0xfu));
    if (threadIdx.x < 64) {
        bool same = false;
        if (rp_wait(xline, status, nb, 1u)) {
            const unsigned v = rp_ld(xline + (threadIdx.x < (unsigned)nb ? threadIdx.x : 0));
            same = __all(v == __builtin_amdgcn_readfirstlane(v));
        }
        if (threadIdx.x == 0) *s_local = same ? 1 : 0;
    }
}
__device__ __forceinline__ void rp_store_frag(const __amdgpu_buffer_rsrc_t rs, int f, int lane, u32x4_t v, bool local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (f * 64 + lane) * 16, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, rs, (f * 64 + lane) * 16, 0, RP_SC1);
}
__device__ __forceinline__ void rp_raise(unsigned* flag, unsigned value, bool local) {        // one lane, after the wave's vmcnt(0)
    if (local) __hip_atomic_store((gu32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store: lands in the shared L2
    else rp_st(flag, value);
}

// `s_waitcnt vmcnt(N)` with N a run-time count of the stores issued BEHIND the hand-off stores (loads, stores and atomics of a wave complete in
// issue order: MI355X_MICROARCH.md, cycle constants): the hand-off tile is out, the younger row-major stores may still be in flight
#define RP_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void rp_wait_all_but(int n) {
    switch (n) {
        RP_VMC(0) RP_VMC(1) RP_VMC(2) RP_VMC(3) RP_VMC(4) RP_VMC(5) RP_VMC(6) RP_VMC(7) RP_VMC(8) RP_VMC(9) RP_VMC(10) RP_VMC(11) RP_VMC(12)
        RP_VMC(13) RP_VMC(14) RP_VMC(15) RP_VMC(16) RP_VMC(17) RP_VMC(18) RP_VMC(19) RP_VMC(20) RP_VMC(21) RP_VMC(22) RP_VMC(23) RP_VMC(24)
        RP_VMC(25) RP_VMC(26) RP_VMC(27) RP_VMC(28) RP_VMC(29) RP_VMC(30) RP_VMC(31) RP_VMC(32) RP_VMC(33) RP_VMC(34) RP_VMC(35) RP_VMC(36)
        RP_VMC(37) RP_VMC(38) RP_VMC(39) RP_VMC(40) RP_VMC(41) RP_VMC(42) RP_VMC(43) RP_VMC(44) RP_VMC(45) RP_VMC(46) RP_VMC(47) RP_VMC(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// ---- wave pairs (the *2 kernels below): two waves of a workgroup share one (row tile, unit tile) and exchange through LDS ----
// LDS serves one wave's instructions in order and is one memory for the whole workgroup, so "data words, then the step word" by the writer and
// "step word, then data words" by the reader needs no hardware fence; RP_LDS_FENCE keeps the compiler from moving the accesses across each other.
__device__ __forceinline__ unsigned rp_lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rp_lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait (bounded) until the LDS word reaches `need`; false: the launch is aborting
__device__ __forceinline__ bool rp_pair_wait(const unsigned* word, unsigned need, unsigned* status) {
    if (rp_lds_ld(word) >= need) return true;
    const long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        if (rp_lds_ld(word) >= need) return true;
        if ((spins & 255u) == 0u) {
            if (rp_ld(status) != 0u) return false;
            if (wall_clock64() - t0 > RP_LIMIT) {
                if ((threadIdx.x & 63) == 0) { rp_st(status, 1u); rp_st(status + 1, 1u); }
                return false;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// forward:  z = xproj[t] (gate-minor, bias included) + h[t-1] . Wh^T ;  i,g,f,o ; c ; h
// One loop iteration of a wave:  wait for its row tile's flags | request the state tile (all k-steps at once) | xproj[t] into the accumulators |
// MFMA chain, weights from LDS one k-step ahead | gate pointwise in registers | hand-off stores, then the row-major stores (gates, c, h, y,
// h^T, y^T: 28 KB per tile) | wait for the hand-off stores only, raise the flag | request the next step's xproj and keep bytes.
// ------------------------------------------------------------------------------------------------------------------
template <int U, bool XB, typename F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_rowpar_fwd_kernel(RFwdArgs A) {
    constexpr int KS = U / 16;                      // k-steps of 16 over the recurrent width
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);     // [gate][k-step][lane] 16-byte B fragments
    int* s_local = reinterpret_cast<int*>(smem + (size_t)KS * 4096 + 4 * 5120);
    const int grp = blockIdx.x % A.G, nt = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int nb = U / 32;
    unsigned* status = A.sync;
    rp_probe_xcd(A.sync + RP_FLAGS_OFF + 32 * nrt + 32 * grp, status, nt, nb, s_local);
    const int rot = (2 * nt) & (KS - 1);           // rotation of this member's k-step order (speed only: the sum over k is taken in that order)
    {   // this workgroup's 128 gate-interleaved rows of wh_t [4U, U] -> LDS, once, k-steps in the member's rotated order
        const int n0 = nt * 128;
#pragma unroll 4
        for (int i = 0; i < KS; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            const int g = e / (KS * 64), s = (e >> 6) % KS, ln = e & 63;
            wl[e] = *reinterpret_cast<const uint4*>(A.wh_t + (size_t)(n0 + 32 * g + (ln & 31)) * U + 16 * ((s + rot) & (KS - 1)) + 8 * (ln >> 5));
        }
    }
    __syncthreads();
    const bool local = A.allow_local && *s_local != 0;
    const int rt = grp + A.G * w;                   // this wave's row tile, for the whole sequence
    if (rt >= nrt) return;
    bf16_t (*sH)[40] = reinterpret_cast<bf16_t (*)[40]>(smem + (size_t)KS * 4096 + (size_t)w * 5120);          // tile [row][unit] (+pad)
    bf16_t (*sT)[40] = reinterpret_cast<bf16_t (*)[40]>(smem + (size_t)KS * 4096 + (size_t)w * 5120 + 2560);   // tile [unit][row]
    unsigned* flags = A.sync + RP_FLAGS_OFF + rt * 32;
    const int m0 = rt * 32, unit = nt * 32 + r;
    const size_t us = (size_t)B * U, slab = (size_t)KS * 1024;
    const bool drop = A.mask != nullptr;
    const float ikp = 1.0f / A.kp;
    // per-lane byte offsets inside the (t, row tile) block of each array; row of register k = rp_krow(k) + 4 hh
    const unsigned og = (unsigned)((4 * hh) * 4 * U + unit * 4) * 4u;         // gates / xproj (gate-minor float4)
    const unsigned oc = (unsigned)((4 * hh) * U + unit) * 4u;                 // c (float)
    const unsigned om = (unsigned)((4 * hh) * U + unit);                      // keep mask (byte)
    const int prow = lane >> 1, pp = (lane & 1) * 2;                          // tile stores: 32 rows x 4 pieces of 16 bytes, two pieces per lane
    f32x16_t acc[4];
    typename RpX<XB, F>::T xv[16];                     // the next step's xproj: requested behind the flag store, moved into acc behind the next wait
    float creg[16];
    unsigned mk[16];
    uint4 mq = make_uint4(0u, 0u, 0u, 0u);          // the keep bytes of the next step: 16 of row (lane >> 1), units 16 (lane & 1) .. + 15
#pragma unroll
    for (int k = 0; k < 16; ++k) { creg[k] = 0.f; mk[k] = 0u; }
    // xproj[t] and the keep bytes into VGPRs, nothing else: the accumulators live in AGPRs, and a copy there (v_accvgpr_write) needs the loaded
    // value -- written here it made every request wait for its own HBM round trip (1.5 us per step in the stage trace).  The keep bytes
    // come as ONE 16-byte load per lane and reach their (row, unit) lanes through the LDS tile at the top of the next item: a wave can
    // have 63 vector-memory operations outstanding (vmcnt is 6 bits), and 40 stores + 16 + 16 loads made every request wait for stores
    const uint8_t* __restrict__ mask_or_c = drop ? A.mask : reinterpret_cast<const uint8_t*>(A.c);
    auto prefetch = [&](int t) {
        constexpr int XS = XB ? 2 : 4;              // bytes per stored value
        const char* xb = reinterpret_cast<const char*>(A.xproj) + ((size_t)t * 4 * us + (size_t)m0 * 4 * U) * XS + og / (4 / XS);
#pragma unroll
        for (int k = 0; k < 16; ++k) xv[k] = *reinterpret_cast<const typename RpX<XB, F>::T*>(xb + (size_t)rp_krow(k) * 4 * XS * U);
        mq = *reinterpret_cast<const uint4*>(mask_or_c + (size_t)t * us + (size_t)(m0 + prow) * U + nt * 32 + (lane & 1) * 16);     // unconditional: see the pair kernels
    };
    prefetch(0);
    const bool trc = nt == 0 && rt == 0;
    for (int t = 0; t < T; ++t) {
        RP_TR(0, trc, t, 0);
        if (t > 0 && !rp_wait(flags, status, nb, (unsigned)t)) return;
        RP_TR(0, trc, t, 1);
        typename F::x8 a[KS];
        {
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(t > 0 ? A.hx + ((size_t)(t - 1) * nrt + rt) * slab : A.hx0 + (size_t)rt * slab, slab);
#pragma unroll
            for (int s = 0; s < KS; ++s)        // k-step s of this member is slab k-step (s + rot) % KS: the members of a row tile start at different lines
                a[s] = __builtin_bit_cast(typename F::x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ((((s + rot) & (KS - 1)) * 64 + lane) * 16), 0, RP_SC1));
        }
        // all KS loads are in flight before anything else: left alone, the scheduler sinks every load next to its use and waits vmcnt(0) per MFMA
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float4 x4 = RpX<XB, F>::get(xv[k]); acc[0][k] = x4.x; acc[1][k] = x4.y; acc[2][k] = x4.z; acc[3][k] = x4.w; }   // z starts at xproj[t]
        if (drop) {                                 // keep bytes: [row][32 units] through the (idle) h tile
            uint8_t* mt = reinterpret_cast<uint8_t*>(&sH[0][0]);
            RP_LDS_FENCE();
            *reinterpret_cast<uint4*>(mt + prow * 32 + (lane & 1) * 16) = mq;
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 16; ++k) mk[k] = mt[(rp_krow(k) + 4 * hh) * 32 + r];
            RP_LDS_FENCE();
        }
        // B fragments (weights in LDS) one k-step AHEAD of their MFMAs: an LDS read takes ~100 cycles to return, an MFMA issues every 32
        typename F::x8 bq[2][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[0][g] = __builtin_bit_cast(typename F::x8, wl[(g * KS + 0) * 64 + lane]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) bq[(s + 1) & 1][g] = __builtin_bit_cast(typename F::x8, wl[(g * KS + s + 1) * 64 + lane]);
            }
            __builtin_amdgcn_sched_barrier(0);                        // ... and keep those reads in front of this k-step's MFMAs
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = F::mfma32(a[s], bq[s & 1][g], acc[g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        RP_TR(0, trc, t, 2);
        // gate pointwise in registers; the activations replace the pre-activations in acc (they are what gets saved)
        unsigned yb[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float gi = fast_sigmoid(acc[0][k]), gg = fast_tanh(acc[1][k]), gf = fast_sigmoid(acc[2][k]), go = fast_sigmoid(acc[3][k]);
            const float cv = gg * gi + creg[k] * gf;
            const float hv = fast_tanh(cv) * go;
            acc[0][k] = gi; acc[1][k] = gg; acc[2][k] = gf; acc[3][k] = go;
            creg[k] = cv;
            const bf16_t hb = F::cvt(hv);
            const int lr = rp_krow(k) + 4 * hh;
            sH[lr][r] = hb;
            sT[r][lr] = hb;
            yb[k] = drop ? (unsigned)F::cvt(F::f32(hb) * ikp * (float)mk[k]) : (unsigned)hb;
        }
        RP_LDS_FENCE();
        RP_TR(0, trc, t, 3);
        {   // hand-off: k-steps 2 nt, 2 nt + 1 of the (t, row tile) slab, A-fragment order (row = lane & 31, 8 units per lane half)
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(A.hx + ((size_t)t * nrt + rt) * slab, slab);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) rp_store_frag(rs, 2 * nt + ks, lane, *reinterpret_cast<const u32x4_t*>(&sH[r][ks * 16 + hh * 8]), local);
        }
        RP_HANDOFF_FENCE();
        RP_TR(0, trc, t, 4);
        // ---- issued BEHIND the hand-off stores, which alone the flag waits for: the next step's operands (first: they are needed soonest),
        // then what the rest of the train step reads ----
        int behind = 0;
        if (t + 1 < T) { prefetch(t + 1); behind += 17; }    // 16 xproj loads + the keep-byte load, which is issued with or without a mask (see prefetch)
        if (A.gates != nullptr) {                 // saved activations: bf16, gate-minor (8 bytes per (row, unit)) -- only mnn_lstm_rowpar_bwd reads them
            char* gb = reinterpret_cast<char*>(A.gates + (size_t)t * 4 * us + (size_t)m0 * 4 * U) + og / 2;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint2 pk;
                pk.x = (unsigned)F::cvt(acc[0][k]) | ((unsigned)F::cvt(acc[1][k]) << 16);
                pk.y = (unsigned)F::cvt(acc[2][k]) | ((unsigned)F::cvt(acc[3][k]) << 16);
                *reinterpret_cast<uint2*>(gb + (size_t)rp_krow(k) * 8 * U) = pk;
            }
            behind += 16;
        }
        if (!drop || t + 1 == T) {   // with dropout the layer's output is y, and only the final state reads h
            bf16_t* dst = A.h + (size_t)t * us + (size_t)(m0 + prow) * U + nt * 32 + pp * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sH[prow][pp * 8]);
            *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(&sH[prow][pp * 8 + 8]);
            behind += 2;
        }
        if (A.hT != nullptr && t + 1 < T) {                          // hT[unit][(t+1) B + row]: the recurrent weight gradient's operand
            bf16_t* dst = A.hT + (size_t)(nt * 32 + prow) * A.ld_hT + (size_t)(t + 1) * B + m0 + pp * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[prow][pp * 8]);
            *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(&sT[prow][pp * 8 + 8]);
            behind += 2;
        }
        if (drop) {                                                   // the dropped output: the same two tiles again
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int lr = rp_krow(k) + 4 * hh;
                sH[lr][r] = (bf16_t)yb[k];
                sT[r][lr] = (bf16_t)yb[k];
            }
            RP_LDS_FENCE();
            bf16_t* dst = A.y + (size_t)t * us + (size_t)(m0 + prow) * U + nt * 32 + pp * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sH[prow][pp * 8]);
            *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(&sH[prow][pp * 8 + 8]);
            behind += 2;
        }
        if (A.yT != nullptr) {                                        // yT[unit][t B + row] (y, or h without dropout)
            bf16_t* dst = A.yT + (size_t)(nt * 32 + prow) * A.ld_yT + (size_t)t * B + m0 + pp * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[prow][pp * 8]);
            *reinterpret_cast<uint4*>(dst + 8) = *reinterpret_cast<const uint4*>(&sT[prow][pp * 8 + 8]);
            behind += 2;
        }
        RP_LDS_FENCE();
        RP_TR(0, trc, t, 5);
        rp_wait_all_but(behind);                                      // the hand-off tile is out
        if (lane == 0) rp_raise(flags + nt, (unsigned)(t + 1), local);
        RP_TR(0, trc, t, 6);
        {
            char* cb = reinterpret_cast<char*>(A.c + (size_t)t * us + (size_t)m0 * U) + oc;
#pragma unroll
            for (int k = 0; k < 16; ++k) *reinterpret_cast<float*>(cb + (size_t)rp_krow(k) * 4 * U) = creg[k];
        }
        RP_TR(0, trc, t, 7);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward:  dh = dh_ext[t] (/ kp * keep when a mask is given) + dz[t+1] . Wh ;  gate backward -> dz[t], dc
// ------------------------------------------------------------------------------------------------------------------
struct RBwdArgs {
    const float* dh_ext; const bf16_t* wh_p; const bf16_t* gates; const float* c; const uint8_t* mask;
    bf16_t* dzc; bf16_t* dzT; int ld_t; float* db_p;
    char* dzx; const char* dzx0; unsigned* sync;
    int T, B, nrt, G, allow_local; float kp;
};

template <int U, typename F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_rowpar_bwd_kernel(RBwdArgs A) {
    constexpr int KS4 = U / 4;                      // k-steps of 16 over the 4U gate columns
    constexpr int CH = 8;                           // k-steps per register chunk of the dz tile (ring of RP_RING chunks)
    constexpr int NCH = KS4 / CH;
    constexpr int MAXW = U == 512 ? 3 : 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);     // [k-step][lane] 16-byte B fragments of wh_p rows nt*32 .. nt*32+31
    int* s_local = reinterpret_cast<int*>(smem + (size_t)KS4 * 1024 + (size_t)MAXW * 10240);
    const int grp = blockIdx.x % A.G, nt = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int nb = U / 32;
    unsigned* status = A.sync;
    rp_probe_xcd(A.sync + RP_FLAGS_OFF + 32 * nrt + 32 * grp, status, nt, nb, s_local);
    const int rot = (8 * nt) & (KS4 - 1);          // rotation of this member's k-step order, in whole chunks (speed only)
#pragma unroll 4
    for (int i = 0; i < KS4 / 4; ++i) {
        const int e = i * 256 + (int)threadIdx.x;
        const int s = e >> 6, ln = e & 63;
        wl[e] = *reinterpret_cast<const uint4*>(A.wh_p + (size_t)(nt * 32 + (ln & 31)) * 4 * U + 16 * ((s + rot) & (KS4 - 1)) + 8 * (ln >> 5));
    }
    __syncthreads();
    const bool local = A.allow_local && *s_local != 0;
    const int rt = grp + A.G * w;
    if (rt >= nrt) return;
    char* tile = smem + (size_t)KS4 * 1024 + (size_t)w * 10240;
    bf16_t (*sZ)[136] = reinterpret_cast<bf16_t (*)[136]>(tile);               // dz tile [row][gate*32 + unit] (+pad): 8704 bytes
    bf16_t (*sT)[32][40] = reinterpret_cast<bf16_t (*)[32][40]>(tile);         // dz tile [gate][unit][row]: 10240 bytes, AFTER sZ has been read
    unsigned* flags = A.sync + RP_FLAGS_OFF + rt * 32;
    const int m0 = rt * 32, unit = nt * 32 + r;
    const size_t us = (size_t)B * U, slab = (size_t)KS4 * 1024;
    const bool drop = A.mask != nullptr;
    const float ikp = 1.0f / A.kp;
    const unsigned og = (unsigned)((4 * hh) * 4 * U + unit * 4) * 4u;
    const unsigned oc = (unsigned)((4 * hh) * U + unit) * 4u;
    const unsigned om = (unsigned)((4 * hh) * U + unit);
    float dcreg[16], cnext[16], dbv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) dcreg[k] = 0.f;
    {   // c[T-1]: from then on an item's c is the previous item's c_prev
        const char* cb = reinterpret_cast<const char*>(A.c + (size_t)(T - 1) * us + (size_t)m0 * U) + oc;
#pragma unroll
        for (int k = 0; k < 16; ++k) cnext[k] = *reinterpret_cast<const float*>(cb + (size_t)rp_krow(k) * 4 * U);
    }
    // epilogue operands of one (t, row tile): gates, c[t-1], the external gradient and the keep bytes.  Loaded one item AHEAD (issued
    // right behind the flag store of the item before), so that they have landed when the hand-off of the next item arrives
    // gates: one 16-byte load per (row, unit).  c[t-1], dh_ext and the keep bytes: whole 16-byte pieces of their 32 x 32 tiles (row lane >> 3
    // + 8 j, units 4 (lane & 7) ..; keep bytes: row lane >> 1, units 16 (lane & 1) ..), 9 loads instead of 48 -- a wave can have 63 vector-memory
    // operations outstanding (vmcnt is 6 bits) and 16 stores + 64 loads made every request wait for the stores; the pieces reach their (row, unit)
    // lanes through the (idle) tile buffer at the top of the next item
    uint2 gv[16];                                   // saved gate activations, bf16 x 4 (i | g << 16, f | o << 16)
    float4 cq[4], dq[4];
    uint4 mq = make_uint4(0u, 0u, 0u, 0u);
    float cp[16];
    const int srow = lane >> 3, spc = lane & 7, prow = lane >> 1;
    const uint8_t* __restrict__ mask_or_c = drop ? A.mask : reinterpret_cast<const uint8_t*>(A.c);
    auto prefetch = [&](int t) {
        const float* pb = A.c + (size_t)(t > 0 ? t - 1 : 0) * us + (size_t)(m0 + srow) * U + nt * 32 + spc * 4;
        const float* db = A.dh_ext + (size_t)t * us + (size_t)(m0 + srow) * U + nt * 32 + spc * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cq[j] = *reinterpret_cast<const float4*>(pb + (size_t)(8 * j) * U);
            dq[j] = *reinterpret_cast<const float4*>(db + (size_t)(8 * j) * U);
        }
        mq = *reinterpret_cast<const uint4*>(mask_or_c + (size_t)t * us + (size_t)(m0 + prow) * U + nt * 32 + (lane & 1) * 16);     // unconditional: see the pair kernels
        const char* gb = reinterpret_cast<const char*>(A.gates + (size_t)t * 4 * us + (size_t)m0 * 4 * U) + og / 2;
#pragma unroll
        for (int k = 0; k < 16; ++k) gv[k] = *reinterpret_cast<const uint2*>(gb + (size_t)rp_krow(k) * 8 * U);
    };
    prefetch(T - 1);
    const bool trc = nt == 0 && rt == 0;
    for (int kk = 0; kk < T; ++kk) {
        const int t = T - 1 - kk;
        RP_TR(1, trc, kk, 0);
        if (kk > 0 && !rp_wait(flags, status, nb, (unsigned)kk)) return;
        RP_TR(1, trc, kk, 1);
        f32x16_t acc;
        {
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(kk > 0 ? A.dzx + ((size_t)(t + 1) * nrt + rt) * slab : A.dzx0 + (size_t)rt * slab, slab);
            typename F::x8 a[RP_RING][CH];            // ring of chunks of eight k-steps (8 KiB each), all but one in flight
            auto issue = [&](int ch) {
#pragma unroll
                for (int s = 0; s < CH; ++s)
                    a[ch % RP_RING][s] = __builtin_bit_cast(typename F::x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ((((ch * CH + s + rot) & (KS4 - 1)) * 64 + lane) * 16), 0, RP_SC1));
            };
#pragma unroll
            for (int ch = 0; ch < RP_RING - 1 && ch < NCH; ++ch) issue(ch);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            // B fragments (weights in LDS) one chunk-half (four k-steps) AHEAD of their MFMAs
            typename F::x8 bq[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[0][q] = __builtin_bit_cast(typename F::x8, wl[q * 64 + lane]);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + RP_RING - 1 < NCH) issue(ch + RP_RING - 1);     // the rest of the ring is in flight while this chunk feeds the MFMAs
                __builtin_amdgcn_sched_barrier(0);           // (left alone, the scheduler sinks each load to its use and waits per MFMA)
#pragma unroll
                for (int s4 = 0; s4 < CH / 4; ++s4) {
                    const int f4 = ch * (CH / 4) + s4;       // flattened group of four k-steps
                    if (f4 + 1 < KS4 / 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[(f4 + 1) & 1][q] = __builtin_bit_cast(typename F::x8, wl[((f4 + 1) * 4 + q) * 64 + lane]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = F::mfma32(a[ch % RP_RING][s4 * 4 + q], bq[f4 & 1][q], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        RP_TR(1, trc, kk, 2);
        float dhx[16];
        {   // the item's staged operands -> their (row, unit) lanes, through the (idle) tile buffer: [32][33] f32 tiles of c[t-1] and dh_ext,
            // [32][32] keep bytes.  Nothing before this point needed them: they were requested a whole item ago
            float* st_c = reinterpret_cast<float*>(tile);
            float* st_d = reinterpret_cast<float*>(tile + 128 * RP_STG);
            uint8_t* st_m = reinterpret_cast<uint8_t*>(tile + 256 * RP_STG);
            RP_LDS_FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) {               // component stores: whole-float4 stores of cq / dq send the arrays to scratch memory
                float* pc_ = st_c + (srow + 8 * j) * RP_STG + spc * 4;
                float* pd_ = st_d + (srow + 8 * j) * RP_STG + spc * 4;
                pc_[0] = cq[j].x; pc_[1] = cq[j].y; pc_[2] = cq[j].z; pc_[3] = cq[j].w;
                pd_[0] = dq[j].x; pd_[1] = dq[j].y; pd_[2] = dq[j].z; pd_[3] = dq[j].w;
            }
            if (drop) *reinterpret_cast<uint4*>(st_m + prow * 32 + (lane & 1) * 16) = mq;
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int lr = rp_krow(k) + 4 * hh;
                cp[k] = st_c[lr * RP_STG + r];
                const float dv = st_d[lr * RP_STG + r];
                dhx[k] = drop ? dv * ikp * (float)st_m[lr * 32 + r] : dv;       // dropout backward of rnn.py:132 on the external gradient
            }
            RP_LDS_FENCE();
        }
        // gate backward in registers (rnn.py:124 LSTMBlockCell autodiff): dz = {d i, d g, d f, d o} pre-activation gradients
        unsigned bvp[16][2];                     // the four bf16 values of a register row, packed (i | g << 16, f | o << 16)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float gi = F::lo(gv[k].x), gg = F::hi(gv[k].x);
            const float gf = F::lo(gv[k].y), go = F::hi(gv[k].y);
            const float dh = dhx[k] + acc[k];
            const float tc = fast_tanh(cnext[k]);
            const float d_o = dh * tc;
            const float d_c = dh * go * (1.f - tc * tc) + dcreg[k];
            const float cprev = t > 0 ? cp[k] : 0.f;
            const float dzv[4] = {d_c * gg * gi * (1.f - gi), d_c * gi * (1.f - gg * gg), d_c * cprev * gf * (1.f - gf), d_o * go * (1.f - go)};
            dcreg[k] = d_c * gf;
            cnext[k] = cprev;
            const int lr = rp_krow(k) + 4 * hh;
            bf16_t b4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                b4[g] = F::cvt(dzv[g]);
                sZ[lr][32 * g + r] = b4[g];
                dbv[g] += F::f32(b4[g]);             // the (bf16) values the weight-gradient GEMMs see
            }
            bvp[k][0] = (unsigned)b4[0] | ((unsigned)b4[1] << 16);
            bvp[k][1] = (unsigned)b4[2] | ((unsigned)b4[3] << 16);
        }
        RP_LDS_FENCE();
        RP_TR(1, trc, kk, 3);
        {   // hand-off: k-steps 8 nt .. 8 nt + 7 of the (t, row tile) slab
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(A.dzx + ((size_t)t * nrt + rt) * slab, slab);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) rp_store_frag(rs, 8 * nt + ks, lane, *reinterpret_cast<const u32x4_t*>(&sZ[r][ks * 16 + hh * 8]), local);
        }
        RP_HANDOFF_FENCE();
        RP_TR(1, trc, kk, 4);
        // ---- issued BEHIND the hand-off stores, which alone the flag waits for: the next item's operands, then what the rest of the step reads ----
        int behind = 0;
        if (kk + 1 < T) { prefetch(t - 1); behind += 25; }
        if (A.dzc != nullptr) {                  // row-major dz [t][row][4U] (gate-interleaved columns): the A operand of the input-gradient GEMM
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = j * 64 + lane, row = p >> 4, pc = p & 15;
                *reinterpret_cast<uint4*>(A.dzc + ((size_t)t * B + m0 + row) * 4 * U + nt * 128 + pc * 8) = *reinterpret_cast<const uint4*>(&sZ[row][pc * 8]);
            }
            behind += 8;
        }
        RP_LDS_FENCE();
        if (A.dzT != nullptr) {                  // dzT[nt*128 + 32 g + unit][t B + row]: the K-contiguous operand of the weight-gradient GEMMs
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int lr = rp_krow(k) + 4 * hh;
                sT[0][r][lr] = (bf16_t)(bvp[k][0] & 0xffffu); sT[1][r][lr] = (bf16_t)(bvp[k][0] >> 16);
                sT[2][r][lr] = (bf16_t)(bvp[k][1] & 0xffffu); sT[3][r][lr] = (bf16_t)(bvp[k][1] >> 16);
            }
            RP_LDS_FENCE();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = j * 64 + lane, gu = p >> 2, piece = p & 3;
                // ld_t == 0: the K-BLOCKED layout [T B / 32][4U][32] (mnn_gemm_tn, MNN_GEMM_A_KBLOCK32): this wave's 128 columns x 32 rows are one
                // contiguous 8 KB slab of block t B / 32 + rt -- every store instruction writes a whole kilobyte
                bf16_t* dst = A.ld_t == 0 ? A.dzT + (((size_t)t * (B >> 5) + rt) * (4 * U) + nt * 128 + gu) * 32 + piece * 8
                                          : A.dzT + (size_t)(nt * 128 + gu) * A.ld_t + (size_t)t * B + m0 + piece * 8;
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&sT[gu >> 5][gu & 31][piece * 8]);
            }
            behind += 8;
        }
        RP_LDS_FENCE();
        RP_TR(1, trc, kk, 5);
        rp_wait_all_but(behind);
        if (lane == 0) rp_raise(flags + nt, (unsigned)(kk + 1), local);
        RP_TR(1, trc, kk, 6);
        RP_TR(1, trc, kk, 7);
    }
    if (A.db_p != nullptr) {                     // bias gradient: this wave's column sums over its rows and all steps
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float s2 = dbv[g] + __shfl_xor(dbv[g], 32);
            if (hh == 0) atomicAdd(A.db_p + nt * 128 + 32 * g + r, s2);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Wave-pair forms (at most two row tiles per workgroup: every (row tile, unit tile) item gets TWO waves, so all four SIMDs of a CU work).
// A single wave per item leaves the step a serial chain on one SIMD: 128 MFMAs (1.7 us at 32 cycles each) + the gate pointwise of 1024
// (row, unit) pairs (1.6 us forward, 2.1 us backward) of the 6.2 / 8.4 us per timestep at U = 512 (profiles/tools/rowpar_trace.py).
//   forward:  the pair splits the GATE columns (wave 0: i, g; wave 1: f, o -- each reads half of the weights in LDS and the whole h[t-1] tile),
//             then the ROWS for the pointwise: wave h keeps C registers 8h .. 8h+7 (rows 16h .. 16h+15), sends the other eight of its two gate
//             tiles to its partner through LDS (4 KiB each way) and receives the partner's gates for its own rows;
//   backward: the pair splits K (wave h: dz columns of unit tiles h nb/2 .. -- half of the weights, half of the 4U-wide dz[t+1] tile: the
//             load stream of a wave is bounded by its 63 outstanding requests), exchanges the partial dh of the other's rows (2 KiB each way).
// Every wave hands on its own 16 rows and raises its own flag (member 2 nt + h): forward consumers wait for all 2 nb flags of the row tile,
// backward consumers for the nb flags of their K half.  Transposed outputs (h^T, y^T, dz^T) leave as 8-byte stores straight from the registers
// (a lane holds four consecutive rows of a unit), no LDS transpose.
// ------------------------------------------------------------------------------------------------------------------
template <int U, bool XB, typename F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_rowpar_fwd2_kernel(RFwdArgs A) {
    constexpr int KS = U / 16;
    constexpr size_t OFF_TILE = (size_t)KS * 4096, TILE_B = 2304, OFF_X = OFF_TILE + 4 * TILE_B, OFF_HS = OFF_X + 2 * 8192;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);     // [gate][k-step][lane] 16-byte B fragments
    unsigned* hs = reinterpret_cast<unsigned*>(smem + OFF_HS);      // [slot][ready 0, ready 1, taken 0, taken 1]
    int* s_local = reinterpret_cast<int*>(smem + OFF_HS + 32);
    const int grp = blockIdx.x % A.G, nt = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = w >> 1, half = w & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int nb = U / 32;
    unsigned* status = A.sync;
    if (threadIdx.x < 8) hs[threadIdx.x] = 0u;
    rp_probe_xcd(A.sync + RP_FLAGS_OFF + 32 * nrt + 32 * grp, status, nt, nb, s_local);
    const int rot = (2 * nt) & (KS - 1);
    {
        const int n0 = nt * 128;
#pragma unroll 4
        for (int i = 0; i < KS; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            const int g = e / (KS * 64), s = (e >> 6) % KS, ln = e & 63;
            wl[e] = *reinterpret_cast<const uint4*>(A.wh_t + (size_t)(n0 + 32 * g + (ln & 31)) * U + 16 * ((s + rot) & (KS - 1)) + 8 * (ln >> 5));
        }
    }
    __syncthreads();
    const bool local = A.allow_local && *s_local != 0;
    const int rt = grp + A.G * slot;                // the pair's row tile, for the whole sequence
    if (rt >= nrt) return;                          // both waves of the pair leave together
    bf16_t (*sH)[40] = reinterpret_cast<bf16_t (*)[40]>(smem + OFF_TILE + (size_t)w * TILE_B);      // this wave's 16 rows [row][unit] (+pad)
    unsigned* sT = reinterpret_cast<unsigned*>(smem + OFF_TILE + (size_t)w * TILE_B + 1280);        // [unit][8 row pairs]: the transposed tile
    float4* xsend = reinterpret_cast<float4*>(smem + OFF_X + (size_t)slot * 8192 + (size_t)half * 4096);
    const float4* xrecv = reinterpret_cast<const float4*>(smem + OFF_X + (size_t)slot * 8192 + (size_t)(half ^ 1) * 4096);
    unsigned* my_ready = hs + slot * 4 + half;
    const unsigned* peer_ready = hs + slot * 4 + (half ^ 1);
    unsigned* my_taken = hs + slot * 4 + 2 + half;          // "I have read the partner's words of step .."
    const unsigned* peer_taken = hs + slot * 4 + 2 + (half ^ 1);
    unsigned* flags = A.sync + RP_FLAGS_OFF + rt * 32;
    const int m0 = rt * 32, unit = nt * 32 + r;
    const int rb = 16 * half + 4 * hh;              // global row (inside the tile) of local register row 0; register kl: rb + rp_krow(kl)
    const size_t us = (size_t)B * U, slab = (size_t)KS * 1024;
    const bool drop = A.mask != nullptr;
    const float ikp = 1.0f / A.kp;
    const unsigned og = (unsigned)(rb * 4 * U + unit * 4) * 4u;           // gates / xproj (gate-minor float4)
    const unsigned oc = (unsigned)(rb * U + unit) * 4u;                   // c (float)
    const int g0 = 2 * half;                        // this wave's two gate tiles: g0, g0 + 1
    f32x16_t acc[2];
    typename RpX<XB, F>::T xv[8];
    float creg[8];
    unsigned mk[8];
    uint4 mq = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int k = 0; k < 8; ++k) { creg[k] = 0.f; mk[k] = 0u; }
    const uint8_t* __restrict__ mask_or_c = drop ? A.mask : reinterpret_cast<const uint8_t*>(A.c);
    auto prefetch = [&](int t) {
        constexpr int XS = XB ? 2 : 4;              // bytes per stored value
        const char* xb = reinterpret_cast<const char*>(A.xproj) + ((size_t)t * 4 * us + (size_t)m0 * 4 * U) * XS + og / (4 / XS);
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = *reinterpret_cast<const typename RpX<XB, F>::T*>(xb + (size_t)rp_krow(k) * 4 * XS * U);
        // (every lane loads -- the upper half repeats the lower half's 16 bytes: under a lane predicate the compiler moved the loaded registers
        // and put an s_waitcnt vmcnt(0) in front of the moves, a full drain of the hand-off stores and prefetches just issued, once per timestep)
        // Unconditional as well (without a mask the same offsets are read from c, which is four times as large, and the result is ignored): a
        // conditionally assigned mq became a phi whose operands sat in different registers -- same moves, same drain.
        mq = *reinterpret_cast<const uint4*>(mask_or_c + (size_t)t * us + (size_t)(m0 + 16 * half + ((lane & 31) >> 1)) * U + nt * 32 + (lane & 1) * 16);
    };
    prefetch(0);
    const bool trc = nt == 0 && rt == 0 && half == 0;
    for (int t = 0; t < T; ++t) {
        RP_TR(0, trc, t, 0);
        // The keep bytes reach their lanes HERE, in front of the flag wait (the time a wave spends waiting anyway): behind the h loads the
        // compiler put s_waitcnt vmcnt(0) in front of this LDS store (the keep bytes were requested a step ago, but the wait it chose also
        // covered the h loads just issued), and the MFMA chain started only when the last of them had landed.
        if (drop) {                                 // keep bytes of this wave's 16 rows through its (idle) h tile
            uint8_t* mt = reinterpret_cast<uint8_t*>(&sH[0][0]);
            RP_LDS_FENCE();
            if (lane < 32) *reinterpret_cast<uint4*>(mt + (lane >> 1) * 32 + (lane & 1) * 16) = mq;
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) mk[k] = mt[(rp_krow(k) + 4 * hh) * 32 + r];
            RP_LDS_FENCE();
        }
        if (t > 0 && !rp_wait(flags, status, 2 * nb, (unsigned)t)) return;
        RP_TR(0, trc, t, 1);
        typename F::x8 a[KS];
        {
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(t > 0 ? A.hx + ((size_t)(t - 1) * nrt + rt) * slab : A.hx0 + (size_t)rt * slab, slab);
            // same-XCD mode: through the CU's L1 -- both waves of the pair (and the second pair) read the same 1-KiB lines within a microsecond,
            // and a slab address is written once per launch before anybody reads it (the L1 starts the launch empty), so a hit is never stale
            if (local) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    a[s] = __builtin_bit_cast(typename F::x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ((((s + rot) & (KS - 1)) * 64 + lane) * 16), 0, RP_A_AUX));
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    a[s] = __builtin_bit_cast(typename F::x8, __builtin_amdgcn_raw_buffer_load_b128(rs, ((((s + rot) & (KS - 1)) * 64 + lane) * 16), 0, RP_SC1));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc[0][k] = 0.f; acc[1][k] = 0.f; }
        typename F::x8 bq[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g) bq[0][g] = __builtin_bit_cast(typename F::x8, wl[((g0 + g) * KS + 0) * 64 + lane]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
                for (int g = 0; g < 2; ++g) bq[(s + 1) & 1][g] = __builtin_bit_cast(typename F::x8, wl[((g0 + g) * KS + s + 1) * 64 + lane]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[g] = F::mfma32(a[s], bq[s & 1][g], acc[g]);
        }
        __builtin_amdgcn_sched_barrier(0);
        RP_TR(0, trc, t, 2);
        // ---- pair exchange: the other wave's rows of my two gate tiles go out, its two gate tiles of my rows come in ----
        float keep[2][8], recv[2][8];
        {
            if (t > 0 && !rp_pair_wait(peer_taken, (unsigned)t, status)) return;      // the partner has read my words of step t - 1
            RP_LDS_FENCE();
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float4 o4;
                    o4.x = half ? acc[g][4 * q + 0] : acc[g][8 + 4 * q + 0];
                    o4.y = half ? acc[g][4 * q + 1] : acc[g][8 + 4 * q + 1];
                    o4.z = half ? acc[g][4 * q + 2] : acc[g][8 + 4 * q + 2];
                    o4.w = half ? acc[g][4 * q + 3] : acc[g][8 + 4 * q + 3];
                    xsend[(g * 2 + q) * 64 + lane] = o4;
                }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int k = 0; k < 8; ++k) keep[g][k] = half ? acc[g][8 + k] : acc[g][k];
            RP_LDS_FENCE();
            if (lane == 0) rp_lds_st(my_ready, (unsigned)(t + 1));
            if (!rp_pair_wait(peer_ready, (unsigned)(t + 1), status)) return;
            RP_LDS_FENCE();
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 i4 = xrecv[(g * 2 + q) * 64 + lane];
                    recv[g][4 * q + 0] = i4.x; recv[g][4 * q + 1] = i4.y; recv[g][4 * q + 2] = i4.z; recv[g][4 * q + 3] = i4.w;
                }
            RP_LDS_FENCE();
            if (lane == 0) rp_lds_st(my_taken, (unsigned)(t + 1));
        }
        // gate pointwise of this wave's 16 rows, in registers
        float gact[4][8];
        unsigned hbv[8], yb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 x4 = RpX<XB, F>::get(xv[k]);
            const float zi = (half ? recv[0][k] : keep[0][k]) + x4.x, zg = (half ? recv[1][k] : keep[1][k]) + x4.y;
            const float zf = (half ? keep[0][k] : recv[0][k]) + x4.z, zo = (half ? keep[1][k] : recv[1][k]) + x4.w;
            const float gi = fast_sigmoid(zi), gg = fast_tanh(zg), gf = fast_sigmoid(zf), go = fast_sigmoid(zo);
            const float cv = gg * gi + creg[k] * gf;
            const float hv = fast_tanh(cv) * go;
            gact[0][k] = gi; gact[1][k] = gg; gact[2][k] = gf; gact[3][k] = go;
            creg[k] = cv;
            const bf16_t hb = F::cvt(hv);
            sH[rp_krow(k) + 4 * hh][r] = hb;
            hbv[k] = (unsigned)hb;
            yb[k] = drop ? (unsigned)F::cvt(F::f32(hb) * ikp * (float)mk[k]) : (unsigned)hb;
        }
        RP_LDS_FENCE();
        RP_TR(0, trc, t, 3);
        {   // hand-off of this wave's 16 rows: k-steps 2 nt, 2 nt + 1 of the slab in ONE store (lane -> k-step, unit half, row)
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(A.hx + ((size_t)t * nrt + rt) * slab, slab);
            const int ks = lane >> 5, kh = (lane >> 4) & 1, rl = lane & 15;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&sH[rl][ks * 16 + kh * 8]);
            const int off = (((2 * nt + ks) * 64) + kh * 32 + 16 * half + rl) * 16;
            if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, RP_SC1);
        }
        RP_HANDOFF_FENCE();
        RP_TR(0, trc, t, 4);
        // Behind the hand-off store: the next step's operands, then everything the row-major outputs need EXCEPT their stores (packing, the
        // LDS transposes) -- that work runs while the hand-off store is on its way; the flag goes up as soon as that ONE store is out, and
        // only then are the row-major stores issued (in front of the flag their issue alone was 0.6 us of every timestep's chain).
        int behind = 0;
        if (t + 1 < T) { prefetch(t + 1); behind += 9; }
        uint2 gpk[8];
        if (A.gates != nullptr) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                gpk[k].x = (unsigned)F::cvt(gact[0][k]) | ((unsigned)F::cvt(gact[1][k]) << 16);
                gpk[k].y = (unsigned)F::cvt(gact[2][k]) | ((unsigned)F::cvt(gact[3][k]) << 16);
            }
        }
        const bool want_h = !drop || t + 1 == T;      // with dropout the layer's output is y, and only the final state reads h
        const bool want_hT = A.hT != nullptr && t + 1 < T;
        uint4 p_h = make_uint4(0u, 0u, 0u, 0u), p_hT = p_h, p_y = p_h, p_yT = p_h;
        if (want_h) p_h = *reinterpret_cast<const uint4*>(&sH[lane >> 2][(lane & 3) * 8]);       // h rows: 16 rows x 4 pieces of 16 bytes
        if (want_hT) {                                // hT[unit][(t+1) B + row]: 16 rows = 32 bytes per unit, through the transposed tile
            RP_LDS_FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) sT[r * 8 + ((rp_krow(2 * j) + 4 * hh) >> 1)] = hbv[2 * j] | (hbv[2 * j + 1] << 16);
            RP_LDS_FENCE();
            p_hT = *reinterpret_cast<const uint4*>(sT + (lane >> 1) * 8 + (lane & 1) * 4);
        }
        if (drop) {
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) sH[rp_krow(k) + 4 * hh][r] = (bf16_t)yb[k];
            RP_LDS_FENCE();
            p_y = *reinterpret_cast<const uint4*>(&sH[lane >> 2][(lane & 3) * 8]);
        }
        if (A.yT != nullptr) {
            RP_LDS_FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) sT[r * 8 + ((rp_krow(2 * j) + 4 * hh) >> 1)] = yb[2 * j] | (yb[2 * j + 1] << 16);
            RP_LDS_FENCE();
            p_yT = *reinterpret_cast<const uint4*>(sT + (lane >> 1) * 8 + (lane & 1) * 4);
        }
        RP_LDS_FENCE();
        RP_TR(0, trc, t, 5);
        rp_wait_all_but(behind);
        if (lane == 0) rp_raise(flags + 2 * nt + half, (unsigned)(t + 1), local);
        RP_TR(0, trc, t, 6);
        if (A.gates != nullptr) {
            char* gb = reinterpret_cast<char*>(A.gates + (size_t)t * 4 * us + (size_t)m0 * 4 * U) + og / 2;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<uint2*>(gb + (size_t)rp_krow(k) * 8 * U) = gpk[k];
        }
        if (want_h) *reinterpret_cast<uint4*>(A.h + (size_t)t * us + (size_t)(m0 + 16 * half + (lane >> 2)) * U + nt * 32 + (lane & 3) * 8) = p_h;
        if (want_hT) *reinterpret_cast<uint4*>(A.hT + (size_t)(nt * 32 + (lane >> 1)) * A.ld_hT + (size_t)(t + 1) * B + m0 + 16 * half + (lane & 1) * 8) = p_hT;
        if (drop) *reinterpret_cast<uint4*>(A.y + (size_t)t * us + (size_t)(m0 + 16 * half + (lane >> 2)) * U + nt * 32 + (lane & 3) * 8) = p_y;
        if (A.yT != nullptr) *reinterpret_cast<uint4*>(A.yT + (size_t)(nt * 32 + (lane >> 1)) * A.ld_yT + (size_t)t * B + m0 + 16 * half + (lane & 1) * 8) = p_yT;
        {
            char* cb = reinterpret_cast<char*>(A.c + (size_t)t * us + (size_t)m0 * U) + oc;
#pragma unroll
            for (int k = 0; k < 8; ++k) *reinterpret_cast<float*>(cb + (size_t)rp_krow(k) * 4 * U) = creg[k];
        }
        RP_TR(0, trc, t, 7);
    }
}

template <int U, typename F>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) lstm_rowpar_bwd2_kernel(RBwdArgs A) {
    constexpr int KS4 = U / 4, HK = KS4 / 2;        // k-steps of 16 over the 4U gate columns; per wave of the pair
    constexpr int CH = 8, NCH = HK / CH;
    constexpr size_t OFF_TILE = (size_t)KS4 * 1024, TILE_B = 4864, OFF_X = OFF_TILE + 4 * TILE_B, OFF_HS = OFF_X + 2 * 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* wl = reinterpret_cast<uint4*>(smem);     // [position][lane] 16-byte B fragments of wh_p rows nt*32 .. +31, each half of K in its walk order
    unsigned* hs = reinterpret_cast<unsigned*>(smem + OFF_HS);
    int* s_local = reinterpret_cast<int*>(smem + OFF_HS + 32);
    const int grp = blockIdx.x % A.G, nt = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = w >> 1, half = w & 1;
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int nb = U / 32;
    unsigned* status = A.sync;
    if (threadIdx.x < 8) hs[threadIdx.x] = 0u;
    rp_probe_xcd(A.sync + RP_FLAGS_OFF + 32 * nrt + 32 * grp, status, nt, nb, s_local);
    const int rot = (8 * nt) & (HK - 1);            // rotation of the walk inside a K half, in whole chunks (speed only)
#pragma unroll 4
    for (int i = 0; i < KS4 / 4; ++i) {
        const int e = i * 256 + (int)threadIdx.x;
        const int p = e >> 6, ln = e & 63;
        const int ks = (p / HK) * HK + ((p % HK + rot) & (HK - 1));
        wl[e] = *reinterpret_cast<const uint4*>(A.wh_p + (size_t)(nt * 32 + (ln & 31)) * 4 * U + 16 * ks + 8 * (ln >> 5));
    }
    __syncthreads();
    const bool local = A.allow_local && *s_local != 0;
    const int rt = grp + A.G * slot;
    if (rt >= nrt) return;
    char* tile = smem + OFF_TILE + (size_t)w * TILE_B;
    bf16_t (*sZ)[136] = reinterpret_cast<bf16_t (*)[136]>(tile);               // dz of this wave's 16 rows [row][gate*32 + unit] (+pad): 4352 bytes
    float4* xsend = reinterpret_cast<float4*>(smem + OFF_X + (size_t)slot * 4096 + (size_t)half * 2048);
    const float4* xrecv = reinterpret_cast<const float4*>(smem + OFF_X + (size_t)slot * 4096 + (size_t)(half ^ 1) * 2048);
    unsigned* my_ready = hs + slot * 4 + half;
    const unsigned* peer_ready = hs + slot * 4 + (half ^ 1);
    unsigned* my_taken = hs + slot * 4 + 2 + half;
    const unsigned* peer_taken = hs + slot * 4 + 2 + (half ^ 1);
    unsigned* flags = A.sync + RP_FLAGS_OFF + rt * 32;
    const int m0 = rt * 32, unit = nt * 32 + r;
    const int rb = 16 * half + 4 * hh;
    const size_t us = (size_t)B * U, slab = (size_t)KS4 * 1024;
    const bool drop = A.mask != nullptr;
    const float ikp = 1.0f / A.kp;
    const unsigned og = (unsigned)(rb * 4 * U + unit * 4) * 4u;
    const unsigned oc = (unsigned)(rb * U + unit) * 4u;
    float dcreg[8], cnext[8], dbv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) dcreg[k] = 0.f;
    {
        const char* cb = reinterpret_cast<const char*>(A.c + (size_t)(T - 1) * us + (size_t)m0 * U) + oc;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnext[k] = *reinterpret_cast<const float*>(cb + (size_t)rp_krow(k) * 4 * U);
    }
    uint2 gv[8];
    float4 cq[2], dq[2];
    uint4 mq = make_uint4(0u, 0u, 0u, 0u);
    float cp[8];
    const int srow = lane >> 3, spc = lane & 7;
    const uint8_t* __restrict__ mask_or_c = drop ? A.mask : reinterpret_cast<const uint8_t*>(A.c);
    auto prefetch = [&](int t) {
        const float* pb = A.c + (size_t)(t > 0 ? t - 1 : 0) * us + (size_t)(m0 + 16 * half + srow) * U + nt * 32 + spc * 4;
        const float* db = A.dh_ext + (size_t)t * us + (size_t)(m0 + 16 * half + srow) * U + nt * 32 + spc * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            cq[j] = *reinterpret_cast<const float4*>(pb + (size_t)(8 * j) * U);
            dq[j] = *reinterpret_cast<const float4*>(db + (size_t)(8 * j) * U);
        }
        // (every lane loads -- the upper half repeats the lower half's 16 bytes: under a lane predicate the compiler moved the loaded registers
        // and put an s_waitcnt vmcnt(0) in front of the moves, a full drain of the hand-off stores and prefetches just issued, once per timestep)
        // Unconditional as well (without a mask the same offsets are read from c, which is four times as large, and the result is ignored): a
        // conditionally assigned mq became a phi whose operands sat in different registers -- same moves, same drain.
        mq = *reinterpret_cast<const uint4*>(mask_or_c + (size_t)t * us + (size_t)(m0 + 16 * half + ((lane & 31) >> 1)) * U + nt * 32 + (lane & 1) * 16);
        const char* gb = reinterpret_cast<const char*>(A.gates + (size_t)t * 4 * us + (size_t)m0 * 4 * U) + og / 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) gv[k] = *reinterpret_cast<const uint2*>(gb + (size_t)rp_krow(k) * 8 * U);
    };
    prefetch(T - 1);
    const bool trc = nt == 0 && rt == 0 && half == 0;
    for (int kk = 0; kk < T; ++kk) {
        const int t = T - 1 - kk;
        RP_TR(1, trc, kk, 0);
        if (kk > 0 && !rp_wait(flags + half * nb, status, nb, (unsigned)kk)) return;      // the producers of this wave's K half (both row halves)
        RP_TR(1, trc, kk, 1);
        f32x16_t acc;
        {
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(kk > 0 ? A.dzx + ((size_t)(t + 1) * nrt + rt) * slab : A.dzx0 + (size_t)rt * slab, slab);
            typename F::x8 a[RP_RING][CH];
            auto issue = [&](int ch) {
#pragma unroll
                for (int s = 0; s < CH; ++s)
                    a[ch % RP_RING][s] = __builtin_bit_cast(typename F::x8, __builtin_amdgcn_raw_buffer_load_b128(
                        rs, (((half * HK + ((ch * CH + s + rot) & (HK - 1))) * 64 + lane) * 16), 0, RP_SC1));
            };
#pragma unroll
            for (int ch = 0; ch < RP_RING - 1 && ch < NCH; ++ch) issue(ch);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            const uint4* wlh = wl + (size_t)half * HK * 64;
            typename F::x8 bq[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[0][q] = __builtin_bit_cast(typename F::x8, wlh[q * 64 + lane]);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + RP_RING - 1 < NCH) issue(ch + RP_RING - 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < CH / 4; ++s4) {
                    const int f4 = ch * (CH / 4) + s4;
                    if (f4 + 1 < HK / 4) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[(f4 + 1) & 1][q] = __builtin_bit_cast(typename F::x8, wlh[((f4 + 1) * 4 + q) * 64 + lane]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = F::mfma32(a[ch % RP_RING][s4 * 4 + q], bq[f4 & 1][q], acc);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        RP_TR(1, trc, kk, 2);
        // ---- pair exchange: my partial dh of the partner's rows goes out, its partial of my rows comes in ----
        float dhp[8];
        {
            if (kk > 0 && !rp_pair_wait(peer_taken, (unsigned)kk, status)) return;
            RP_LDS_FENCE();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float4 o4;
                o4.x = half ? acc[4 * q + 0] : acc[8 + 4 * q + 0];
                o4.y = half ? acc[4 * q + 1] : acc[8 + 4 * q + 1];
                o4.z = half ? acc[4 * q + 2] : acc[8 + 4 * q + 2];
                o4.w = half ? acc[4 * q + 3] : acc[8 + 4 * q + 3];
                xsend[q * 64 + lane] = o4;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dhp[k] = half ? acc[8 + k] : acc[k];
            RP_LDS_FENCE();
            if (lane == 0) rp_lds_st(my_ready, (unsigned)(kk + 1));
            if (!rp_pair_wait(peer_ready, (unsigned)(kk + 1), status)) return;
            RP_LDS_FENCE();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 i4 = xrecv[q * 64 + lane];
                dhp[4 * q + 0] += i4.x; dhp[4 * q + 1] += i4.y; dhp[4 * q + 2] += i4.z; dhp[4 * q + 3] += i4.w;
            }
            RP_LDS_FENCE();
            if (lane == 0) rp_lds_st(my_taken, (unsigned)(kk + 1));
        }
        float dhx[8];
        {   // staged operands of this wave's 16 rows -> their (row, unit) lanes through the (idle) tile buffer
            float* st_c = reinterpret_cast<float*>(tile);
            float* st_d = reinterpret_cast<float*>(tile + 64 * RP_STG);
            uint8_t* st_m = reinterpret_cast<uint8_t*>(tile + 4224);
            RP_LDS_FENCE();
#pragma unroll
            for (int j = 0; j < 2; ++j) {               // component stores: whole-float4 stores of cq / dq send the arrays to scratch memory
                float* pc_ = st_c + (srow + 8 * j) * RP_STG + spc * 4;
                float* pd_ = st_d + (srow + 8 * j) * RP_STG + spc * 4;
                pc_[0] = cq[j].x; pc_[1] = cq[j].y; pc_[2] = cq[j].z; pc_[3] = cq[j].w;
                pd_[0] = dq[j].x; pd_[1] = dq[j].y; pd_[2] = dq[j].z; pd_[3] = dq[j].w;
            }
            if (drop && lane < 32) *reinterpret_cast<uint4*>(st_m + (lane >> 1) * 32 + (lane & 1) * 16) = mq;
            RP_LDS_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int lr = rp_krow(k) + 4 * hh;
                cp[k] = st_c[lr * RP_STG + r];
                const float dv = st_d[lr * RP_STG + r];
                dhx[k] = drop ? dv * ikp * (float)st_m[lr * 32 + r] : dv;
            }
            RP_LDS_FENCE();
        }
        unsigned bz[4][8];                        // bf16 dz by gate and register row
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gi = F::lo(gv[k].x), gg = F::hi(gv[k].x);
            const float gf = F::lo(gv[k].y), go = F::hi(gv[k].y);
            const float dh = dhx[k] + dhp[k];
            const float tc = fast_tanh(cnext[k]);
            const float d_o = dh * tc;
            const float d_c = dh * go * (1.f - tc * tc) + dcreg[k];
            const float cprev = t > 0 ? cp[k] : 0.f;
            const float dzv[4] = {d_c * gg * gi * (1.f - gi), d_c * gi * (1.f - gg * gg), d_c * cprev * gf * (1.f - gf), d_o * go * (1.f - go)};
            dcreg[k] = d_c * gf;
            cnext[k] = cprev;
            const int lr = rp_krow(k) + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bf16_t b = F::cvt(dzv[g]);
                sZ[lr][32 * g + r] = b;
                dbv[g] += F::f32(b);
                bz[g][k] = (unsigned)b;
            }
        }
        RP_LDS_FENCE();
        RP_TR(1, trc, kk, 3);
        {   // hand-off of this wave's 16 rows: k-steps 8 nt .. 8 nt + 7 of the slab, two k-steps per store
            const __amdgpu_buffer_rsrc_t rs = rp_rsrc(A.dzx + ((size_t)t * nrt + rt) * slab, slab);
            const int kh = (lane >> 4) & 1, rl = lane & 15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ks = 2 * j + (lane >> 5);
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&sZ[rl][ks * 16 + kh * 8]);
                const int off = (((8 * nt + ks) * 64) + kh * 32 + 16 * half + rl) * 16;
                if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, RP_SC1);
            }
        }
        RP_HANDOFF_FENCE();
        RP_TR(1, trc, kk, 4);
        // As in the forward: the next item's operands and the LDS work of the row-major outputs run while the hand-off stores are on their way,
        // the flag goes up as soon as those four stores are out, the eight row-major stores are issued behind it.
        int behind = 0;
        if (kk + 1 < T) { prefetch(t - 1); behind += 13; }
        // (the pieces are read whether or not their output is wanted: registers defined under one condition and stored under another end up in scratch memory)
        u32x4_t p_c0, p_c1, p_c2, p_c3, p_t0, p_t1, p_t2, p_t3;
        {
            const int p = lane;
            p_c0 = *reinterpret_cast<const u32x4_t*>(&sZ[(p + 0) >> 4][(p & 15) * 8]);
            p_c1 = *reinterpret_cast<const u32x4_t*>(&sZ[(p + 64) >> 4][(p & 15) * 8]);
            p_c2 = *reinterpret_cast<const u32x4_t*>(&sZ[(p + 128) >> 4][(p & 15) * 8]);
            p_c3 = *reinterpret_cast<const u32x4_t*>(&sZ[(p + 192) >> 4][(p & 15) * 8]);
        }
        {                                        // dzT[nt*128 + 32 g + unit][t B + row]: 16 rows = 32 bytes per (gate, unit), through the transposed tile
            unsigned* sT = reinterpret_cast<unsigned*>(tile);        // [gate*32 + unit][8 row pairs], over the dz tile (read above)
            RP_LDS_FENCE();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int j = 0; j < 4; ++j) sT[(g * 32 + r) * 8 + ((rp_krow(2 * j) + 4 * hh) >> 1)] = bz[g][2 * j] | (bz[g][2 * j + 1] << 16);
            RP_LDS_FENCE();
            const int p = lane;
            p_t0 = *reinterpret_cast<const u32x4_t*>(sT + ((p + 0) >> 1) * 8 + (p & 1) * 4);
            p_t1 = *reinterpret_cast<const u32x4_t*>(sT + ((p + 64) >> 1) * 8 + (p & 1) * 4);
            p_t2 = *reinterpret_cast<const u32x4_t*>(sT + ((p + 128) >> 1) * 8 + (p & 1) * 4);
            p_t3 = *reinterpret_cast<const u32x4_t*>(sT + ((p + 192) >> 1) * 8 + (p & 1) * 4);
        }
        RP_LDS_FENCE();
        RP_TR(1, trc, kk, 5);
        rp_wait_all_but(behind);
        if (lane == 0) rp_raise(flags + 2 * nt + half, (unsigned)(kk + 1), local);
        if (A.dzc != nullptr) {
            auto put = [&](int j, const u32x4_t& v) {
                const int p = j * 64 + lane, row = p >> 4, pc = p & 15;
                *reinterpret_cast<u32x4_t*>(A.dzc + ((size_t)t * B + m0 + 16 * half + row) * 4 * U + nt * 128 + pc * 8) = v;
            };
            put(0, p_c0); put(1, p_c1); put(2, p_c2); put(3, p_c3);
        }
        if (A.dzT != nullptr) {
            auto put = [&](int j, const u32x4_t& v) {
                const int p = j * 64 + lane, gu = p >> 1, piece = p & 1;
                // ld_t == 0: the K-BLOCKED layout [T B / 32][4U][32] (see the one-wave kernel): the pair's two halves interleave 32-byte runs
                // inside one contiguous 8 KB slab
                bf16_t* dst = A.ld_t == 0 ? A.dzT + (((size_t)t * (B >> 5) + rt) * (4 * U) + nt * 128 + gu) * 32 + 16 * half + piece * 8
                                          : A.dzT + (size_t)(nt * 128 + gu) * A.ld_t + (size_t)t * B + m0 + 16 * half + piece * 8;
                *reinterpret_cast<u32x4_t*>(dst) = v;
            };
            put(0, p_t0); put(1, p_t1); put(2, p_t2); put(3, p_t3);
        }
        RP_TR(1, trc, kk, 6);
        RP_TR(1, trc, kk, 7);
    }
    if (A.db_p != nullptr) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float s2 = dbv[g] + __shfl_xor(dbv[g], 32);
            if (hh == 0) atomicAdd(A.db_p + nt * 128 + 32 * g + r, s2);
        }
    }
}

// ---------------------------------------------------------------------------------------------- host side
static int rp_cu_count() {          // of the CURRENT device (a process may drive several)
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
static hipError_t rp_prepare_device();
static bool rp_units_ok(int u) { return u == 128 || u == 256 || u == 512; }
// G row-tile groups x U/32 workgroups; every workgroup's waves own one row tile each (at most `maxw`): false when that does not cover B
static bool rp_plan(int B, int U, bool bwd, int& nrt, int& G) {
    if (!rp_units_ok(U) || B <= 0 || (B % 32) != 0) return false;
    const int cus = rp_cu_count(), nb = U / 32;
    if (cus < nb) return false;
    nrt = B / 32;
    G = cus / nb;
    if (G > nrt) G = nrt;
    if (const char* e = getenv("MNN_ROWPAR_MAX_G")) G = std::max(1, std::min(G, atoi(e)));      // test hook: several row tiles per workgroup at small B
    const int tiles = (nrt + G - 1) / G;
    const int maxw = (bwd && U == 512) ? 3 : 4;     // LDS: the backward tile buffers are 10 KiB per wave next to 128 KiB of weights
    return tiles <= maxw;
}
// wave-pair forms: at most two row tiles per workgroup (two waves per item, four waves per workgroup)
static bool rp_pair(int B, int U) {
    static int off = -1;
    if (off < 0) off = getenv("MNN_ROWPAR_NO_PAIR") != nullptr ? 1 : 0;
    int nrt, G;
    if (off || !rp_plan(B, U, false, nrt, G)) return false;
    return (nrt + G - 1) / G <= 2;
}
static size_t rp_sync_bytes(int nrt) { return ((size_t)RP_FLAGS_OFF + 64 * (size_t)nrt) * sizeof(unsigned); }      // flag lines, then one XCC-id line per row-tile group
static size_t rp_edge_bytes(int nrt, int U) { return (size_t)nrt * (size_t)(U / 4) * 1024; }         // zero slabs standing for dz[T] (>= h[-1]'s)
static size_t rp_xchg_off(int nrt, int U) { return (rp_sync_bytes(nrt) + rp_edge_bytes(nrt, U) + 255) / 256 * 256; }

extern "C" int mnn_lstm_rowpar_ok(int B, int units) {
    int nrt, G;
    if (!(rp_plan(B, units, false, nrt, G) && rp_plan(B, units, true, nrt, G))) return 0;
    return rp_prepare_device() == hipSuccess ? 1 : 0;
}
extern "C" size_t mnn_lstm_rowpar_workspace_bytes(int T, int B, int units) {
    const size_t nrt = (size_t)((B + 31) / 32);
    return rp_xchg_off((int)nrt, units) + (size_t)T * nrt * (size_t)(units / 4) * 1024;
}
extern "C" int mnn_lstm_rowpar_status(const void* workspace, int* status) {
    MNN_REQUIRE(workspace && status, "mnn_lstm_rowpar_status: bad arguments");
    unsigned v = 0;
    MNN_HIP(hipMemcpy(&v, (const char*)workspace + sizeof(unsigned), sizeof(v), hipMemcpyDeviceToHost));
    *status = (int)v;
    return MNN_OK;
}

__global__ void rp_reset_kernel(unsigned* sync, int words) {          // progress words back to zero; [1] (sticky) is left alone
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x)
        if (i != 1) sync[i] = 0u;
}

// the workspace layout and the per-launch reset, for the cluster form (lstm_cluster.hip), which takes the same workspace
void mnn_rp_workspace_layout(int nrt, int U, size_t* sync_bytes, size_t* xchg_off) {
    *sync_bytes = rp_sync_bytes(nrt);
    *xchg_off = rp_xchg_off(nrt, U);
}
int mnn_rp_reset_launch(hipStream_t st, void* workspace, int nrt, int U) {
    (void)U;
    hipLaunchKernelGGL(rp_reset_kernel, dim3(8), dim3(256), 0, st, (unsigned*)workspace, (int)(rp_sync_bytes(nrt) / sizeof(unsigned)));
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// Every instantiation with its dynamic-LDS size.  The MaxDynamicSharedMemorySize attribute is per device: it is set for ALL of them the first
// time a device asks mnn_lstm_rowpar_ok / launches (the host calls _ok before every use, so never for the first time under stream capture).
static size_t rp_lds_fwd(int U, bool pair) { return pair ? (size_t)(U / 16) * 4096 + 4 * 2304 + 2 * 8192 + 64 : (size_t)(U / 16) * 4096 + 4 * 5120 + 16; }
static size_t rp_lds_bwd(int U, bool pair) {
    return pair ? (size_t)(U / 4) * 1024 + 4 * 4864 + 2 * 4096 + 64 : (size_t)(U / 4) * 1024 + (size_t)((U == 512) ? 3 : 4) * 10240 + 16;
}
typedef void (*rp_fwd_fn)(RFwdArgs);
typedef void (*rp_bwd_fn)(RBwdArgs);
template <typename F, bool XB> static rp_fwd_fn rp_fwd_kernel_x(int U, bool pair) {
    if (pair) return U == 512 ? lstm_rowpar_fwd2_kernel<512, XB, F> : (U == 256 ? lstm_rowpar_fwd2_kernel<256, XB, F> : lstm_rowpar_fwd2_kernel<128, XB, F>);
    return U == 512 ? lstm_rowpar_fwd_kernel<512, XB, F> : (U == 256 ? lstm_rowpar_fwd_kernel<256, XB, F> : lstm_rowpar_fwd_kernel<128, XB, F>);
}
static rp_fwd_fn rp_fwd_kernel(int U, bool pair, bool xb, bool f16) {
    if (f16) return xb ? rp_fwd_kernel_x<Fp16F, true>(U, pair) : rp_fwd_kernel_x<Fp16F, false>(U, pair);
    return xb ? rp_fwd_kernel_x<Bf16F, true>(U, pair) : rp_fwd_kernel_x<Bf16F, false>(U, pair);
}
template <typename F> static rp_bwd_fn rp_bwd_kernel_x(int U, bool pair) {
    if (pair) return U == 512 ? lstm_rowpar_bwd2_kernel<512, F> : (U == 256 ? lstm_rowpar_bwd2_kernel<256, F> : lstm_rowpar_bwd2_kernel<128, F>);
    return U == 512 ? lstm_rowpar_bwd_kernel<512, F> : (U == 256 ? lstm_rowpar_bwd_kernel<256, F> : lstm_rowpar_bwd_kernel<128, F>);
}
static rp_bwd_fn rp_bwd_kernel(int U, bool pair, bool f16) { return f16 ? rp_bwd_kernel_x<Fp16F>(U, pair) : rp_bwd_kernel_x<Bf16F>(U, pair); }
static hipError_t rp_prepare_device() {
    static bool done[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    const int us[3] = {128, 256, 512};
    for (int ui = 0; ui < 3; ++ui)
        for (int pair = 0; pair < 2; ++pair)
            for (int f16 = 0; f16 < 2; ++f16) {
                for (int xb = 0; xb < 2; ++xb) {
                    e = hipFuncSetAttribute((const void*)rp_fwd_kernel(us[ui], pair, xb, f16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp_lds_fwd(us[ui], pair));
                    if (e != hipSuccess) return e;
                }
                e = hipFuncSetAttribute((const void*)rp_bwd_kernel(us[ui], pair, f16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp_lds_bwd(us[ui], pair));
                if (e != hipSuccess) return e;
            }
    done[dev] = true;
    return hipSuccess;
}

extern "C" int mnn_lstm_rowpar_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L, float keep_prob, void* workspace) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_rowpar_fwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm_rowpar_fwd: workspace must be 256-byte aligned");
    RFwdArgs a{};
    const int U = L->units;
    MNN_REQUIRE(rp_plan(B, U, false, a.nrt, a.G), "mnn_lstm_rowpar_fwd: units must be 128/256/512, B a multiple of 32 and the row tiles must fit the "
                                                  "device's workgroups (B=%d u=%d)", B, U);
    MNN_REQUIRE(L->xproj && L->wh_t && L->c && L->h, "mnn_lstm_rowpar_fwd: null pointer");
    MNN_REQUIRE(L->h0 == nullptr && L->c0 == nullptr, "mnn_lstm_rowpar_fwd: an initial state is not supported by this form (zero state per window)");
    MNN_REQUIRE(L->hT == nullptr || (L->ld_hT >= T * B && (L->ld_hT & 7) == 0), "mnn_lstm_rowpar_fwd: ld_hT too small / not a multiple of 8");
    MNN_REQUIRE(L->yT == nullptr || (L->ld_yT >= T * B && (L->ld_yT & 7) == 0), "mnn_lstm_rowpar_fwd: ld_yT too small / not a multiple of 8");
    MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                "mnn_lstm_rowpar_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    a.xproj = L->xproj; a.wh_t = (const bf16_t*)L->wh_t; a.gates = (bf16_t*)L->gates; a.c = L->c; a.h = (bf16_t*)L->h; a.y = (bf16_t*)L->y; a.mask = L->mask;
    a.hT = (bf16_t*)L->hT; a.ld_hT = L->ld_hT; a.yT = (bf16_t*)L->yT; a.ld_yT = L->ld_yT;
    a.sync = (unsigned*)workspace; a.hx0 = (const char*)workspace + rp_sync_bytes(a.nrt); a.hx = (char*)workspace + rp_xchg_off(a.nrt, U);
    a.T = T; a.B = B; a.kp = keep_prob; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    hipLaunchKernelGGL(rp_reset_kernel, dim3(8), dim3(256), 0, st, (unsigned*)workspace, (int)(rp_sync_bytes(a.nrt) / sizeof(unsigned)));
    MNN_HIP(mnn_zero_async((char*)workspace + rp_sync_bytes(a.nrt), rp_edge_bytes(a.nrt, U), st));
    const int grid = a.G * (U / 32);
    const bool pair = rp_pair(B, U);
    MNN_HIP(rp_prepare_device());
    hipLaunchKernelGGL(rp_fwd_kernel(U, pair, L->xproj_bf16 != 0, L->f16 != 0), dim3(grid), dim3(256), rp_lds_fwd(U, pair), st, a);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_lstm_rowpar_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L, float keep_prob, void* workspace) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm_rowpar_bwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm_rowpar_bwd: workspace must be 256-byte aligned");
    RBwdArgs a{};
    const int U = L->units;
    MNN_REQUIRE(rp_plan(B, U, true, a.nrt, a.G), "mnn_lstm_rowpar_bwd: units must be 128/256/512, B a multiple of 32 and the row tiles must fit the "
                                                 "device's workgroups (B=%d u=%d)", B, U);
    MNN_REQUIRE(L->dh_ext && L->wh_p && L->gates && L->c, "mnn_lstm_rowpar_bwd: null pointer");
    MNN_REQUIRE(L->c0 == nullptr && L->dz == nullptr, "mnn_lstm_rowpar_bwd: no initial state / f32 dz output in this form");
    MNN_REQUIRE(L->dzT_t == nullptr || L->ld_t == 0 || (L->ld_t >= T * B && (L->ld_t & 7) == 0),
                "mnn_lstm_rowpar_bwd: ld_t too small / not a multiple of 8 (0 = the K-blocked layout [T*B/32][4u][32])");
    a.dh_ext = L->dh_ext; a.wh_p = (const bf16_t*)L->wh_p; a.gates = (const bf16_t*)L->gates; a.c = L->c; a.mask = L->mask;
    a.dzc = (bf16_t*)L->dz_T; a.dzT = (bf16_t*)L->dzT_t; a.ld_t = L->ld_t; a.db_p = L->db_p;
    a.sync = (unsigned*)workspace; a.dzx0 = (const char*)workspace + rp_sync_bytes(a.nrt); a.dzx = (char*)workspace + rp_xchg_off(a.nrt, U);
    a.T = T; a.B = B; a.kp = keep_prob; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    hipLaunchKernelGGL(rp_reset_kernel, dim3(8), dim3(256), 0, st, (unsigned*)workspace, (int)(rp_sync_bytes(a.nrt) / sizeof(unsigned)));
    MNN_HIP(mnn_zero_async((char*)workspace + rp_sync_bytes(a.nrt), rp_edge_bytes(a.nrt, U), st));
    const int grid = a.G * (U / 32);
    const bool pair = rp_pair(B, U);
    MNN_HIP(rp_prepare_device());
    hipLaunchKernelGGL(rp_bwd_kernel(U, pair, L->f16 != 0), dim3(grid), dim3(256), rp_lds_bwd(U, pair), st, a);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
