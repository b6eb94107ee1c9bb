// Persistent two-layer LSTM recurrence for gfx950 (bf16 operands, f32 state).
//
// The per-timestep launches of gemm.hip re-read every weight from L2/MALL at each step and pay a kernel boundary per
// step; the recurrence is a latency chain, so both sit on the critical path.  Here ONE launch runs all T steps:
//   * a workgroup owns one (unit tile, row tile) of one layer for the whole sequence and keeps its slice of the
//     recurrent weights in REGISTERS (32 units x K, K split over the waves) -- weights are read once per launch;
//   * the 32-row hidden-state tiles move between workgroups through an EXCHANGE area in global memory, laid out in MFMA
//     A-fragment order ([t][row tile][k-step][lane][8 bf16]) so that every hand-off load and store instruction moves one
//     contiguous KiB: 16-byte write-through (sc1) stores, `s_waitcnt vmcnt(0)` in every storing wave, workgroup
//     barrier, one sc1 flag store per workgroup; consumers poll the flags of their row tile with sc1 loads from one wave,
//     join a barrier, then read with 16-byte sc1 buffer loads (MI355X_MICROARCH.md, "Valid forms", first table row;
//     cdna_hip_programming.md G16 R1).  The row-major copies the rest of the train step needs are written with plain
//     stores one item later, off the chain (after the next hand-off's loads have been issued);
//   * only workgroups of the same ROW TILE ever wait for each other (16 + 8 of them for units 512/256), never the grid;
//   * layer 2 consumes layer 1's step t as soon as its flags say so: its input projection is folded into its step
//     (K = U1 + U2), so the xproj round trip of the launch-per-step form disappears too.
// Every spin is bounded (1 s of the 100 MHz realtime counter) and watches a status word: a workgroup that gives up
// sets it and every other one leaves at its next poll, so the grid always drains.  The host entry refuses shapes whose
// grid is not resident at once (one workgroup per CU).  Block -> tile map: blocks with equal (id % G) share a row tile,
// so with G = 8 a row tile's workgroups share an XCD under round-robin dispatch (speed only, never correctness).
#include "common.h"
#include <stdlib.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

#ifdef PST_TRACE     // development only: per-stage timestamps of one workgroup per role (profiles/tools/persist_trace.hip reads them)
__device__ long long pst_trace[4][512][8];
#define PST_TR(ptr, k) do { if (ptr) (ptr)[k] = wall_clock64(); } while (0)
#define PST_TRP(role, cond, step) ((cond) && threadIdx.x == 0 ? &pst_trace[role][step][0] : nullptr)
#else
#define PST_TR(ptr, k) do { } while (0)
#define PST_TRP(role, cond, step) nullptr
#endif
#define PST_LIMIT 100000000LL      // spin bound: 1 s of wall_clock64() (100 MHz)
#define PST_FLAGS_OFF 32           // words: [0] status, [32 + 32*rt + member] progress flags, then one sticky word
#define PST_SC1 16                 // aux bit of raw buffer loads/stores: device-scope (write-through / L1-bypassing)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// k-steps [f0, f0+KS) of one exchange slab ([k-step][lane][16 B]): one contiguous KiB per instruction
template <int KS, typename V>
__device__ __forceinline__ void load_frags_xchg(const void* slab, size_t slab_bytes, int f0, V (&f)[KS]) {
    const __amdgpu_buffer_rsrc_t rs = slice_rsrc(slab, slab_bytes);
    const int off = (f0 * 64 + (int)(threadIdx.x & 63)) * 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 1024 * s, 0, PST_SC1));
}
// `local`: every workgroup of this row tile sits on ONE XCD (checked at launch start, pst_same_xcd): then the tile is stored
// write-BACK -- it stays in that XCD's L2, which all of them share, and the consumers' sc1 (L1-bypassing, L2-served) loads hit
// it there instead of fetching a write-through line back through the fabric.  Otherwise write-through (device scope).
__device__ __forceinline__ void store_frag_xchg(void* slab, size_t slab_bytes, int f, int lane, u32x4_t v, bool local) {
    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, slice_rsrc(slab, slab_bytes), (f * 64 + lane) * 16, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(v, slice_rsrc(slab, slab_bytes), (f * 64 + lane) * 16, 0, PST_SC1);
}
template <int KS, typename V>
__device__ __forceinline__ void load_frags_plain(const bf16_t* __restrict__ p, V (&f)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const V*>(p + 16 * s);
}

// Workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt(0), which would make
// every barrier of a step wait for the prefetch loads (HBM latency) and deferred stores issued just before it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave 0 polls one 128-byte line of progress flags: lanes [0,n1) need >= need1, lanes [n1,nm) need >= need2, the rest
// watch the status word.  All threads of the workgroup call this; returns false (uniformly) when the launch is aborting.
// `peek` (optional): this lane's word of the SAME line read ahead of time by pst_peek -- flags only grow, so an early observation that
// already satisfies the wait is as good as a fresh one and saves the poll's round trip (~0.35 us) at the top of the item.
__device__ __forceinline__ unsigned pst_peek(const unsigned* line, const unsigned* status, int nm) {
    const int lane = threadIdx.x & 63;                  // every wave reads (unconditional load: no wait is forced behind it); wave 0 uses it
    return ld_agent(lane < nm ? line + lane : status);
}
__device__ __forceinline__ bool pst_wait(const unsigned* line, unsigned* status, int n1, unsigned need1, int nm, unsigned need2, int* s_abort,
                                         int sticky_off, bool have_peek = false, unsigned peek = 0u) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool mine = lane < nm;
        const unsigned need = lane < n1 ? need1 : need2;
        const unsigned* p = mine ? line + lane : status;
        const long long t0 = wall_clock64();
        for (unsigned spins = 1;; ++spins) {
            const unsigned v = (have_peek && spins == 1u) ? peek : ld_agent(p);
            if (__all(mine ? v >= need : v == 0u)) break;
            const bool dead = __any(!mine && v != 0u) || ((spins & 127u) == 0u && wall_clock64() - t0 > PST_LIMIT);
            if (dead) {
                if (lane == 0) { st_agent(status, 1u); st_agent(status + sticky_off, 1u); *s_abort = 1; }   // sticky: survives the next launch's re-zeroing
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    lds_barrier();
    return *reinterpret_cast<volatile int*>(s_abort) == 0;
}

// Publish: every storing wave drains its write-through stores, the workgroup meets, one lane raises the flag.
__device__ __forceinline__ void pst_publish(unsigned* flag, unsigned value, bool local) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (local) __hip_atomic_store((gu32*)flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // plain store: lands in the shared L2
        else st_agent(flag, value);
    }
}

// Launch start: do the nm workgroups of this row-tile group share an XCD?  Each posts 0x100 | XCC_ID (device scope), waits for
// the others (bounded) and compares.  The answer only selects the store policy of the hand-offs; results never depend on it.
__device__ __forceinline__ bool pst_same_xcd(unsigned* xline, unsigned* status, int member, int nm, int* s_abort, int* s_local, int sticky_off) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) st_agent(xline + member, 0x100u | (xcc & 0xfu));
    if (!pst_wait(xline, status, nm, 1u, nm, 1u, s_abort, sticky_off)) return false;
    if (threadIdx.x < 64) {
        const unsigned v = ld_agent(xline + (threadIdx.x < (unsigned)nm ? threadIdx.x : 0));
        const unsigned v0 = __builtin_amdgcn_readfirstlane(v);
        const bool same = __all(v == v0);
        if (threadIdx.x == 0) *s_local = same ? 1 : 0;
    }
    __syncthreads();
    return true;
}

struct PFwdLayer {
    const float* xproj; const bf16_t* wh_t; const float* c0;
    float* gates; float* c; bf16_t* h; bf16_t* hT; int ld_hT; bf16_t* y; const uint8_t* mask;
    const bf16_t* wx_t; int ld_w; const float* bias_p; int U;
    bf16_t* yT; int ld_yT;         // transposed copy of the layer's output (y, or h without dropout): yT[unit][t B + row]
    char* hx; char* yx;            // exchange copies of h[t] (and of the dropped output y[t]), A-fragment order, slab (t, row tile)
    const char* hx0;               // slabs of the initial state h[-1] (zeros, or h0 re-laid by pst_fill_h0_kernel), one per row tile
};
struct PFwdArgs { PFwdLayer l1, l2; int T, B, nrt, G, R; float kp; unsigned* sync; int xcc_off, allow_local, sticky_off; };

struct FwdTiles {
    __attribute__((aligned(16))) float red[4][4][16][64];       // K-split partial tiles, addressed as float4 [producer wave][gate][consumer wave][lane]
    bf16_t sH[32][40];             // h tile [row][unit] (+pad)
    bf16_t sY[32][40];             // dropped output tile
    bf16_t sT[32][40];             // h tile [unit][row] for the transposed copy (weight-gradient operand)
    bf16_t sYT[32][40];            // output tile [unit][row] (y, or h without dropout) for the transposed copy
    int abort, local;
};

// What one finished item leaves for its deferred tail (plain stores issued one item later, off the chain).
struct FwdTail { float gv[4][4], cv[4]; int t, m0; bool valid; };

// Gate pointwise of this wave's 4 fragment rows -> tiles in LDS, then the hand-off of the h (and y) tile.
template <typename F>
__device__ __forceinline__ void pf_finish(const PFwdLayer& L, FwdTiles& S, int T, int nrt, int t, int rt, int nt, float kp, const float (&z)[4][4],
                                          const float (&cp)[4], const unsigned (&mk)[4], FwdTail& tl, unsigned* flag, bool local, long long* trc) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const bool drop = L.mask != nullptr;
    const bool wantT = L.hT != nullptr && t + 1 < T;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float gi = fast_sigmoid(z[q][0]), gg = fast_tanh(z[q][1]), gf = fast_sigmoid(z[q][2]), go = fast_sigmoid(z[q][3]);
        const float c = gg * gi + cp[q] * gf;
        const float h = fast_tanh(c) * go;
        tl.gv[q][0] = gi; tl.gv[q][1] = gg; tl.gv[q][2] = gf; tl.gv[q][3] = go; tl.cv[q] = c;
        const bf16_t hb = F::cvt(h);
        const int lr = 8 * w + q + 4 * hh;
        S.sH[lr][r] = hb;
        bf16_t yb = hb;
        if (drop) { yb = F::cvt(F::f32(hb) / kp * (float)mk[q]); S.sY[lr][r] = yb; }     // mk: raw keep byte, converted here (not at the load)
        if (wantT) S.sT[r][lr] = hb;
        if (L.yT != nullptr) S.sYT[r][lr] = yb;
    }
    tl.t = t; tl.m0 = rt * 32; tl.valid = true;
    lds_barrier();
    PST_TR(trc, 3);
    {   // exchange slab of (t, row tile): k-steps 2nt, 2nt+1 of this unit tile; waves 0,1 store h, waves 2,3 the dropped y
        const size_t slab = (size_t)(L.U / 16) * 1024;
        const bool second = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) != 0;
        const int ks = (threadIdx.x >> 6) & 1;
        if (!second || (drop && L.yx != nullptr)) {
            const bf16_t (*src)[40] = second ? S.sY : S.sH;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&src[r][ks * 16 + hh * 8]);
            store_frag_xchg((second ? L.yx : L.hx) + ((size_t)t * nrt + rt) * slab, slab, 2 * nt + ks, lane, v, local);
        }
    }
    pst_publish(flag, (unsigned)(t + 1), local);
    PST_TR(trc, 4);
}

// Deferred plain stores of a finished item: gates, c, the row-major h / y tiles, the transposed h tile.
__device__ __forceinline__ void pf_tail(const PFwdLayer& L, const FwdTiles& S, int T, int B, int nt, const FwdTail& tl) {
    if (!tl.valid) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int U = L.U, N4 = 4 * U, n0 = nt * 128, unit = nt * 32 + r, t = tl.t, m0 = tl.m0;
    const size_t us = (size_t)B * U;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = m0 + 8 * w + q + 4 * hh;
        if (row >= B) continue;
        const size_t uo = (size_t)t * us + (size_t)row * U + unit;
        if (L.gates != nullptr)                       // gate-minor: the four gates of a (row, unit) are one 16-byte store
            *reinterpret_cast<float4*>(L.gates + (size_t)t * 4 * us + (size_t)row * N4 + unit * 4) = make_float4(tl.gv[q][0], tl.gv[q][1], tl.gv[q][2], tl.gv[q][3]);
        L.c[uo] = tl.cv[q];
    }
    {   // row-major tiles: 32 rows x 64 bytes, one 16-byte store per thread (h: threads 0..127, y: 128..255)
        const int tt = threadIdx.x & 127, row = tt >> 2, piece = tt & 3;
        const bool second = threadIdx.x >= 128;
        if (m0 + row < B && (!second || L.mask != nullptr)) {
            bf16_t* dst = (second ? L.y : L.h) + (size_t)t * us + (size_t)(m0 + row) * U + nt * 32 + piece * 8;
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(second ? &S.sY[row][piece * 8] : &S.sH[row][piece * 8]);
        }
    }
    if (L.yT != nullptr && threadIdx.x >= 128) {                  // yT[unit][t B + row]
        const int tt = threadIdx.x - 128, uu = tt >> 2, piece = tt & 3;
        const int row = m0 + piece * 8, colT = t * B;
        bf16_t* dst = L.yT + (size_t)(nt * 32 + uu) * L.ld_yT + colT + row;
        if (row + 8 <= B && (((size_t)(colT + row) & 7) == 0) && ((L.ld_yT & 7) == 0)) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&S.sYT[uu][piece * 8]);
        } else {
            for (int k = 0; k < 8; ++k)
                if (row + k < B) dst[k] = S.sYT[uu][piece * 8 + k];
        }
    }
    if (L.hT != nullptr && t + 1 < T && threadIdx.x < 128) {      // hT[unit][(t+1) B + row]
        const int uu = threadIdx.x >> 2, piece = threadIdx.x & 3;
        const int row = m0 + piece * 8, colT = (t + 1) * B;
        bf16_t* dst = L.hT + (size_t)(nt * 32 + uu) * L.ld_hT + colT + row;
        if (row + 8 <= B && (((size_t)(colT + row) & 7) == 0) && ((L.ld_hT & 7) == 0)) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&S.sT[uu][piece * 8]);
        } else {
            for (int k = 0; k < 8; ++k)
                if (row + k < B) dst[k] = S.sT[uu][piece * 8 + k];
        }
    }
}

// Per-thread BYTE offsets of this thread's epilogue / tail accesses inside a 32-row tile, computed once per launch.  Every access of an
// item is then  (wave-uniform 64-bit base of the (t, row tile))  +  (one of these 32-bit offsets): scalar address arithmetic and a
// `global_* v_off, s[base]` instruction.  Forming `(size_t)t * 4 * B * U + (size_t)row * 4U + unit * 4` per lane and per access, under
// per-row `row < B` masks, was ~20 instructions and three branches per store: the deferred stores of an item cost 1.0 us of the 4.4 us
// step (profiles/tools/persist_trace.hip with the tail removed: 3.34 us).
struct FwdLane {
    unsigned og[4], oc[4];         // gate-minor float4 (gates, xproj) / per-unit float (c) of fragment rows q = 0..3, full tile
    unsigned ogl[4], ocl[4];       // the same with rows clamped to the LAST row tile (when B is not a multiple of 32)
    unsigned om[4], oml[4];        // keep-mask bytes
    unsigned oh, oyT, ohT;         // 16-byte pieces of the row-major h / y tile and of the transposed tiles
    bool al8;                      // B, ld_yT, ld_hT multiples of 8: the 16-byte tile stores are aligned
};
__device__ __forceinline__ void fwd_lane_init(FwdLane& fl, const PFwdLayer& L, int B, int nrt, int nt) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5, U = L.U, unit = nt * 32 + r;
    const int last_rows = B - (nrt - 1) * 32;               // rows of the last tile (1..32)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = 8 * w + q + 4 * hh, rowl = min(row, last_rows - 1);
        fl.og[q] = (unsigned)(row * 4 * U + unit * 4) * 4u;  fl.ogl[q] = (unsigned)(rowl * 4 * U + unit * 4) * 4u;
        fl.oc[q] = (unsigned)(row * U + unit) * 4u;          fl.ocl[q] = (unsigned)(rowl * U + unit) * 4u;
        fl.om[q] = (unsigned)(row * U + unit);               fl.oml[q] = (unsigned)(rowl * U + unit);
    }
    const int tt = threadIdx.x & 127;
    fl.oh = (unsigned)((tt >> 2) * U + (tt & 3) * 8) * 2u;
    fl.oyT = (unsigned)((tt >> 2) * L.ld_yT + (tt & 3) * 8) * 2u;
    fl.ohT = (unsigned)((tt >> 2) * L.ld_hT + (tt & 3) * 8) * 2u;
    fl.al8 = (B & 7) == 0 && (L.ld_yT & 7) == 0 && (L.ld_hT & 7) == 0;
}
// Deferred plain stores of a finished item when its row tile is full and everything is 16-byte aligned: no per-row predicate, uniform bases.
__device__ __forceinline__ void pf_tail_fast(const PFwdLayer& L, const FwdTiles& S, int T, int B, int nt, const FwdTail& tl, const FwdLane& fl) {
    const int U = L.U, t = tl.t, m0 = tl.m0;
    const size_t us = (size_t)B * U;
    if (L.gates != nullptr) {                                // uniform
        char* gb = reinterpret_cast<char*>(L.gates + (size_t)t * 4 * us + (size_t)m0 * 4 * U);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(gb + fl.og[q]) = make_float4(tl.gv[q][0], tl.gv[q][1], tl.gv[q][2], tl.gv[q][3]);
    }
    {
        char* cb = reinterpret_cast<char*>(L.c + (size_t)t * us + (size_t)m0 * U);
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float*>(cb + fl.oc[q]) = tl.cv[q];
    }
    const bool second = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) != 0;     // waves 2, 3
    const int tt = threadIdx.x & 127, row = tt >> 2, piece = tt & 3;
    if (!second || L.mask != nullptr) {
        char* hb = reinterpret_cast<char*>((second ? L.y : L.h) + (size_t)t * us + (size_t)m0 * U + nt * 32);
        *reinterpret_cast<uint4*>(hb + fl.oh) = *reinterpret_cast<const uint4*>(second ? &S.sY[row][piece * 8] : &S.sH[row][piece * 8]);
    }
    if (second) {
        if (L.yT != nullptr) {                               // yT[unit][t B + row]
            char* yb = reinterpret_cast<char*>(L.yT + (size_t)(nt * 32) * L.ld_yT + (size_t)t * B + m0);
            *reinterpret_cast<uint4*>(yb + fl.oyT) = *reinterpret_cast<const uint4*>(&S.sYT[row][piece * 8]);
        }
    } else if (L.hT != nullptr && t + 1 < T) {               // hT[unit][(t+1) B + row]
        char* tb = reinterpret_cast<char*>(L.hT + (size_t)(nt * 32) * L.ld_hT + (size_t)(t + 1) * B + m0);
        *reinterpret_cast<uint4*>(tb + fl.ohT) = *reinterpret_cast<const uint4*>(&S.sT[row][piece * 8]);
    }
}
__device__ __forceinline__ void pf_tail_any(const PFwdLayer& L, const FwdTiles& S, int T, int B, int nt, const FwdTail& tl, const FwdLane& fl) {
    if (!tl.valid) return;
    if (fl.al8 && tl.m0 + 32 <= B) pf_tail_fast(L, S, T, B, nt, tl, fl);      // uniform
    else pf_tail(L, S, T, B, nt, tl);
}

// Rule for the stretch between a hand-off's loads and its MFMAs: unconditional loads into registers only -- a load
// under a branch feeds a phi, the phi's copy is a use, and the compiler then waits (vmcnt(0)) right behind the load.
// So addresses are selected, never loads; `zero` points at always-zero words of the workspace.

// KS1 = U1 / 64, KS2 = U2 / 64 (k-steps of 16 per wave, 4 waves).
template <int KS1, int KS2, typename F>
__global__ void __launch_bounds__(256) lstm2_persist_fwd_kernel(PFwdArgs A) {
    __shared__ FwdTiles S;
    const int nb1 = A.l1.U / 32, nb2 = A.l2.U / 32, nm = nb1 + nb2;
    const int grp = blockIdx.x % A.G, member = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int Rv = (nrt - grp + A.G - 1) / A.G;            // row tiles of this workgroup: grp, grp + G, ...
    unsigned* status = A.sync;
    unsigned* flags = A.sync + PST_FLAGS_OFF;
    const float* zero = reinterpret_cast<const float*>(A.sync + 4);
    if (threadIdx.x == 0) { S.abort = 0; S.local = 0; }
    __syncthreads();
    if (!pst_same_xcd(A.sync + A.xcc_off + grp * 32, status, member, nm, &S.abort, &S.local, A.sticky_off)) return;
    const bool local = A.allow_local && S.local != 0;
    float4* red4 = reinterpret_cast<float4*>(&S.red[0][0][0][0]);
    FwdTail tl;
    tl.valid = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) tl.cv[q] = 0.f;
    if (member < nb1) {
        // ---------------- layer 1: z = xproj[t] + h[t-1] . Wh^T ----------------
        const PFwdLayer& L = A.l1;
        const int nt = member, U = L.U, N4 = 4 * U, n0 = nt * 128, unit = nt * 32 + r;
        const int kb = w * 16 * KS1 + hh * 8;
        const size_t us = (size_t)B * U, slab = (size_t)(U / 16) * 1024;
        typename F::x8 b[4][KS1];
#pragma unroll
        for (int g = 0; g < 4; ++g) load_frags_plain(L.wh_t + (size_t)(n0 + 32 * g + r) * U + kb, b[g]);
        float xp[4][4];
        unsigned mk[4];
        FwdLane fl;
        fwd_lane_init(fl, L, B, nrt, nt);
        const bool has_mask = L.mask != nullptr;
        auto epi_load = [&](int t, int m0, float (&xp_)[4][4], unsigned (&mk_)[4]) {
            // uniform bases of the (t, row tile) + this thread's offsets (rows clamped in a partial last tile); unconditional loads
            const bool lastp = m0 + 32 > B;
            const char* xb = reinterpret_cast<const char*>(L.xproj + (size_t)t * 4 * us + (size_t)m0 * N4);     // gate-minor xproj
            const uint8_t* mb = has_mask ? L.mask + (size_t)t * us + (size_t)m0 * U : reinterpret_cast<const uint8_t*>(zero);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xv = *reinterpret_cast<const float4*>(xb + (lastp ? fl.ogl[q] : fl.og[q]));
                xp_[q][0] = xv.x; xp_[q][1] = xv.y; xp_[q][2] = xv.z; xp_[q][3] = xv.w;
                mk_[q] = mb[has_mask ? (lastp ? fl.oml[q] : fl.om[q]) : 0u];
            }
        };
        epi_load(0, grp * 32, xp, mk);
        const int n_items = T * Rv;
        bool have_peek = false;
        unsigned peek = 0u;
        for (int i = 0; i < n_items; ++i) {
            const int t = i / Rv, rt = grp + A.G * (i - t * Rv), m0 = rt * 32;
            long long* trc = PST_TRP(0, member == 0 && rt == 0, t);
            PST_TR(trc, 0);
            if (t > 0 && !pst_wait(flags + rt * 32, status, nb1, (unsigned)t, nb1, 0u, &S.abort, A.sticky_off, have_peek, peek)) return;
            PST_TR(trc, 1);
            typename F::x8 a[KS1];
            load_frags_xchg(t > 0 ? L.hx + ((size_t)(t - 1) * nrt + rt) * slab : L.hx0 + (size_t)rt * slab, slab, w * KS1, a);
            // behind the hand-off loads: this item's cell state and the next item's operands (loads: their wait merges with the hand-off's;
            // behind the MFMAs they would be waited for together with the plain stores)
            float cl[4];
            {
                const bool use = t == 0 ? L.c0 != nullptr : Rv != 1;        // uniform: otherwise an always-zero word is read (and not used)
                const char* cb = reinterpret_cast<const char*>(t == 0 ? (L.c0 != nullptr ? L.c0 + (size_t)m0 * U : zero)
                                                                    : (Rv == 1 ? zero : L.c + (size_t)(t - 1) * us + (size_t)m0 * U));
                const bool lastp = m0 + 32 > B;
#pragma unroll
                for (int q = 0; q < 4; ++q) cl[q] = *reinterpret_cast<const float*>(cb + (use ? (lastp ? fl.ocl[q] : fl.oc[q]) : 0u));
            }
            float xpn[4][4];
            unsigned mkn[4];
            {
                const int i2 = min(i + 1, n_items - 1), t2 = i2 / Rv;
                epi_load(t2, (grp + A.G * (i2 - t2 * Rv)) * 32, xpn, mkn);
            }
            f32x16_t acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[g][k] = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = F::mfma32(a[s], b[g][s], acc[g]);
            pf_tail_any(L, S, T, B, nt, tl, fl);           // the previous item's plain stores issue while the MFMA chain runs
            // K-split partials through LDS, 16 bytes per access: slot (producer wave, gate, consumer wave, lane) holds the four accumulator
            // elements 4 w' .. 4 w' + 3 that consumer wave w' reduces (16 stores + 16 loads per thread and item instead of 64 + 64)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int wc = 0; wc < 4; ++wc)
                    red4[((w * 4 + g) * 4 + wc) * 64 + lane] = make_float4(acc[g][4 * wc], acc[g][4 * wc + 1], acc[g][4 * wc + 2], acc[g][4 * wc + 3]);
            lds_barrier();
            float z[4][4], cp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 p0 = red4[((0 * 4 + g) * 4 + w) * 64 + lane], p1 = red4[((1 * 4 + g) * 4 + w) * 64 + lane];
                const float4 p2 = red4[((2 * 4 + g) * 4 + w) * 64 + lane], p3 = red4[((3 * 4 + g) * 4 + w) * 64 + lane];
                z[0][g] = xp[0][g] + ((p0.x + p1.x) + (p2.x + p3.x));
                z[1][g] = xp[1][g] + ((p0.y + p1.y) + (p2.y + p3.y));
                z[2][g] = xp[2][g] + ((p0.z + p1.z) + (p2.z + p3.z));
                z[3][g] = xp[3][g] + ((p0.w + p1.w) + (p2.w + p3.w));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[q] = (Rv == 1 && t > 0) ? tl.cv[q] : cl[q];
            PST_TR(trc, 2);
            {   // several row tiles per workgroup: the next item is another row tile, published a whole round ago -- read its flags now,
                // under the pointwise phase, instead of at the top of the item
                const int i2 = min(i + 1, n_items - 1), t2 = i2 / Rv;
                have_peek = Rv > 1 && i + 1 < n_items;
                peek = pst_peek(flags + (grp + A.G * (i2 - t2 * Rv)) * 32, status, nb1);
            }
            pf_finish<F>(L, S, T, nrt, t, rt, nt, A.kp, z, cp, mk, tl, flags + rt * 32 + member, local, trc);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                mk[q] = mkn[q];
#pragma unroll
                for (int g = 0; g < 4; ++g) xp[q][g] = xpn[q][g];
            }
            PST_TR(trc, 5);
        }
        pf_tail_any(L, S, T, B, nt, tl, fl);
    } else {
        // ---------------- layer 2: z = bias + y1[t] . Wx^T + h[t-1] . Wh^T ----------------
        const PFwdLayer& L = A.l2;
        const PFwdLayer& L1 = A.l1;
        const int nt = member - nb1, U = L.U, U1 = L1.U, n0 = nt * 128, unit = nt * 32 + r;
        const int kb1 = w * 16 * KS1 + hh * 8, kb2 = w * 16 * KS2 + hh * 8;
        const size_t us = (size_t)B * U, slab = (size_t)(U / 16) * 1024, slab1 = (size_t)(U1 / 16) * 1024;
        const char* y1x = L1.mask != nullptr ? L1.yx : L1.hx;
        typename F::x8 bx[4][KS1], bh[4][KS2];
        float bz[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            load_frags_plain(L.wx_t + (size_t)(n0 + 32 * g + r) * L.ld_w + kb1, bx[g]);
            load_frags_plain(L.wh_t + (size_t)(n0 + 32 * g + r) * U + kb2, bh[g]);
            bz[g] = L.bias_p[n0 + 32 * g + r];
        }
        FwdLane fl;
        fwd_lane_init(fl, L, B, nrt, nt);
        const bool has_mask = L.mask != nullptr;
        const int n_items = T * Rv;
        bool have_peek = false;
        unsigned peek = 0u;
        for (int i = 0; i < n_items; ++i) {
            const int t = i / Rv, rt = grp + A.G * (i - t * Rv), m0 = rt * 32;
            long long* trc = PST_TRP(1, member == nb1 && rt == 0, t);
            PST_TR(trc, 0);
            if (!pst_wait(flags + rt * 32, status, nb1, (unsigned)(t + 1), nm, (unsigned)t, &S.abort, A.sticky_off, have_peek, peek)) return;
            PST_TR(trc, 1);
            typename F::x8 a1[KS1], a2[KS2];
            load_frags_xchg(y1x + ((size_t)t * nrt + rt) * slab1, slab1, w * KS1, a1);
            load_frags_xchg(t > 0 ? L.hx + ((size_t)(t - 1) * nrt + rt) * slab : L.hx0 + (size_t)rt * slab, slab, w * KS2, a2);
            float cl[4];
            unsigned mk[4];
            {
                const bool use = t == 0 ? L.c0 != nullptr : Rv != 1;        // uniform (see layer 1)
                const char* cb = reinterpret_cast<const char*>(t == 0 ? (L.c0 != nullptr ? L.c0 + (size_t)m0 * U : zero)
                                                                    : (Rv == 1 ? zero : L.c + (size_t)(t - 1) * us + (size_t)m0 * U));
                const uint8_t* mb = has_mask ? L.mask + (size_t)t * us + (size_t)m0 * U : reinterpret_cast<const uint8_t*>(zero);
                const bool lastp = m0 + 32 > B;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cl[q] = *reinterpret_cast<const float*>(cb + (use ? (lastp ? fl.ocl[q] : fl.oc[q]) : 0u));
                    mk[q] = mb[has_mask ? (lastp ? fl.oml[q] : fl.om[q]) : 0u];
                }
            }
            f32x16_t acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[g][k] = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = F::mfma32(a1[s], bx[g][s], acc[g]);
#pragma unroll
            for (int s = 0; s < KS2; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = F::mfma32(a2[s], bh[g][s], acc[g]);
            pf_tail_any(L, S, T, B, nt, tl, fl);           // the previous item's plain stores issue while the MFMA chain runs
            // K-split partials through LDS, 16 bytes per access: slot (producer wave, gate, consumer wave, lane) holds the four accumulator
            // elements 4 w' .. 4 w' + 3 that consumer wave w' reduces (16 stores + 16 loads per thread and item instead of 64 + 64)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int wc = 0; wc < 4; ++wc)
                    red4[((w * 4 + g) * 4 + wc) * 64 + lane] = make_float4(acc[g][4 * wc], acc[g][4 * wc + 1], acc[g][4 * wc + 2], acc[g][4 * wc + 3]);
            lds_barrier();
            float z[4][4], cp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 p0 = red4[((0 * 4 + g) * 4 + w) * 64 + lane], p1 = red4[((1 * 4 + g) * 4 + w) * 64 + lane];
                const float4 p2 = red4[((2 * 4 + g) * 4 + w) * 64 + lane], p3 = red4[((3 * 4 + g) * 4 + w) * 64 + lane];
                z[0][g] = bz[g] + ((p0.x + p1.x) + (p2.x + p3.x));
                z[1][g] = bz[g] + ((p0.y + p1.y) + (p2.y + p3.y));
                z[2][g] = bz[g] + ((p0.z + p1.z) + (p2.z + p3.z));
                z[3][g] = bz[g] + ((p0.w + p1.w) + (p2.w + p3.w));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[q] = (Rv == 1 && t > 0) ? tl.cv[q] : cl[q];
            PST_TR(trc, 2);
            {   // flags of the next item (another row tile when Rv > 1), read under the pointwise phase
                const int i2 = min(i + 1, n_items - 1), t2 = i2 / Rv;
                have_peek = Rv > 1 && i + 1 < n_items;
                peek = pst_peek(flags + (grp + A.G * (i2 - t2 * Rv)) * 32, status, nm);
            }
            pf_finish<F>(L, S, T, nrt, t, rt, nt, A.kp, z, cp, mk, tl, flags + rt * 32 + member, local, trc);
            PST_TR(trc, 5);
        }
        pf_tail_any(L, S, T, B, nt, tl, fl);
    }
}

// h0 [B,U] row-major -> one exchange slab per row tile (initial state of a stateful call), both layers in one launch (a sampling scan
// makes this call once per generated step); grid (nrt, U1/16 + U2/16), 64 threads; a layer without h0 keeps its zero slabs
__global__ void __launch_bounds__(64) pst_fill_h0_kernel(const bf16_t* __restrict__ h0a, int Ua, char* __restrict__ slab_a,
                                                         const bf16_t* __restrict__ h0b, int Ub, char* __restrict__ slab_b, int B) {
    const bool first = (int)blockIdx.y < Ua / 16;
    const bf16_t* __restrict__ h0 = first ? h0a : h0b;
    if (h0 == nullptr) return;
    const int U = first ? Ua : Ub;
    char* __restrict__ slab0 = first ? slab_a : slab_b;
    const int rt = blockIdx.x, f = first ? blockIdx.y : blockIdx.y - Ua / 16, lane = threadIdx.x;
    const int row = min(rt * 32 + (lane & 31), B - 1);
    const uint4 v = *reinterpret_cast<const uint4*>(h0 + (size_t)row * U + f * 16 + (lane >> 5) * 8);
    *reinterpret_cast<uint4*>(slab0 + ((size_t)rt * (U / 16) + f) * 1024 + lane * 16) = v;
}

// ------------------------------------------------------------------------------------------------------------------
// Backward.  512-thread workgroups (K split over 8 waves), one 32-row x 32-unit tile of dh each:
//   layer 2, step t:  dh = dh_ext[t] + dz2[t+1] . Wh2            (K = 4 U2)
//   layer 1, step t:  dh = (dz2[t] . Wx2) * keep/kp + dz1[t+1] . Wh1   (K = 4 U2 + 4 U1; the dgrad of layer 2's input
//                     projection through the dropout of rnn.py:132 is folded in)
// followed by the gate pointwise; the handed-off tile is dz (bf16, 32 rows x 128 gate-interleaved columns = 8 k-steps of
// its row tile's exchange slab).  No row-major dz is written: only this launch reads it.  dz[T] is a slab of zeros.
// ------------------------------------------------------------------------------------------------------------------
struct PBwdLayer {
    const float* dh_ext; const bf16_t* wh_p; const float* gates; const float* c; const float* c0;
    float* dc; bf16_t* dzTt; int ld_t; const uint8_t* mask; const bf16_t* wx_p; int U;
    float* db_p;                   // bias gradient [4U] (gate-interleaved), accumulated in registers over the launch, or NULL
    char* dzx;                     // exchange copy of dz[t], A-fragment order, slab (t, row tile)
    const char* dzxT;              // zero slabs standing for dz[T], one per row tile
};
struct PBwdArgs { PBwdLayer l1, l2; int T, B, nrt, G, R; float kp; unsigned* sync; int xcc_off, allow_local, sticky_off; };

struct BwdTiles {
    __attribute__((aligned(16))) float red[2][8][16][64];      // addressed as float2 [set][producer wave][consumer wave][lane]
    bf16_t sZ[32][136];            // dz tile [row][gate*32 + unit] (+pad)
    bf16_t sT[4][32][40];          // dz tile [gate][unit][row] for the transposed copy
    int abort, local;
};
struct BwdEpi { float dh, g[4], c, cp; unsigned keep; };      // per fragment row: external gradient, gates, cell states, raw keep byte
struct BwdTail { float dcv[2]; float dbv[4]; int t, m0; bool valid; };   // dbv[g]: this thread's running sum of dz (gate g, its unit, its rows): bias gradient

template <typename F>
__device__ __forceinline__ void pb_finish(const PBwdLayer& L, BwdTiles& S, int B, int nrt, int t, int rt, int nt, const float (&dh)[2], const BwdEpi (&e)[2],
                                          const float (&e_dc)[2], BwdTail& tl, unsigned* flag, unsigned epoch, bool local, long long* trc) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        const int lr = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float gi = e[q].g[0], gg = e[q].g[1], gf = e[q].g[2], go = e[q].g[3];
        const float tc = fast_tanh(e[q].c);
        const float d_o = dh[q] * tc;
        const float d_c = dh[q] * go * (1.f - tc * tc) + e_dc[q];
        const float dzv[4] = {d_c * gg * gi * (1.f - gi), d_c * gi * (1.f - gg * gg), d_c * e[q].cp * gf * (1.f - gf), d_o * go * (1.f - go)};
        tl.dcv[q] = d_c * gf;
        const bool rowok = rt * 32 + lr < B;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16_t bv = F::cvt(dzv[g]);
            S.sZ[lr][32 * g + r] = bv;
            if (L.dzTt != nullptr) S.sT[g][r][lr] = bv;
            tl.dbv[g] += rowok ? F::f32(bv) : 0.f;          // the (bf16) values the weight-gradient GEMMs see, summed in registers
        }
    }
    tl.t = t; tl.m0 = rt * 32; tl.valid = true;
    lds_barrier();
    PST_TR(trc, 3);
    {   // k-steps 8nt .. 8nt+7 of the slab, one per wave: a contiguous KiB per store instruction
        const size_t slab = (size_t)(4 * L.U / 16) * 1024;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&S.sZ[r][w * 16 + hh * 8]);
        store_frag_xchg(L.dzx + ((size_t)t * nrt + rt) * slab, slab, 8 * nt + w, lane, v, local);
    }
    pst_publish(flag, epoch, local);
    PST_TR(trc, 4);
}

__device__ __forceinline__ void pb_tail(const PBwdLayer& L, const BwdTiles& S, int B, int Rv, int nt, BwdTail& tl) {
    if (!tl.valid) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int U = L.U, unit = nt * 32 + r, t = tl.t, m0 = tl.m0;
    if (Rv > 1) {                                   // with one row tile per workgroup the cell gradient never leaves its registers
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 2 * w + q;
            const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (row < B) L.dc[(size_t)row * U + unit] = tl.dcv[q];
        }
    }
    if (L.dzTt != nullptr) {              // dzT_t[pc + 32 g][t B + row]
        const int gu = threadIdx.x >> 2, piece = threadIdx.x & 3;
        const int g = gu >> 5, uu = gu & 31;
        const int row = m0 + piece * 8, colT = t * B;
        bf16_t* dst = L.dzTt + (size_t)(nt * 128 + 32 * g + uu) * L.ld_t + colT + row;
        if (row + 8 <= B && (((size_t)(colT + row) & 7) == 0) && ((L.ld_t & 7) == 0)) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&S.sT[g][uu][piece * 8]);
        } else {
            for (int k = 0; k < 8; ++k)
                if (row + k < B) dst[k] = S.sT[g][uu][piece * 8 + k];
        }
    }
}

// End of the launch: the 16 per-thread partial sums of every gate column (8 waves x 2 lane halves) meet in LDS once, then one f32
// atomic per column (the other row tiles' workgroups add to the same words).  Per item this costs four adds in registers; reading
// the 32 rows of the dz tile back from LDS per item (32 2-byte loads for a quarter of the threads) sat in front of every MFMA chain.
__device__ __forceinline__ void pb_flush_db(const PBwdLayer& L, BwdTiles& S, int nt, const BwdTail& tl) {
    if (L.db_p == nullptr) return;                       // uniform
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    float* part = &S.red[0][0][0][0];                    // [16 (w, hh)][128 columns]: the K-split area is idle by now
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) part[(2 * w + hh) * 128 + 32 * g + r] = tl.dbv[g];
    __syncthreads();
    if (threadIdx.x < 128) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += part[k * 128 + threadIdx.x];
        atomicAdd(L.db_p + nt * 128 + threadIdx.x, sum);
    }
}

// Epilogue operands of one (t, row tile) for this wave's 2 fragment rows: unconditional loads (see the rule above).
template <bool LAYER1>
__device__ __forceinline__ void pb_epi_load(const PBwdLayer& L, int B, int nt, int t, int m0, const float* zero, BwdEpi (&e)[2]) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int U = L.U, N4 = 4 * U, unit = nt * 32 + r, pc = nt * 128 + r;
    const size_t us = (size_t)B * U;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        const int rr = min(m0 + (i & 3) + 8 * (i >> 2) + 4 * hh, B - 1);
        const size_t uo = (size_t)rr * U + unit;
        if (LAYER1) e[q].dh = 0.f;
        else e[q].dh = L.dh_ext[(size_t)t * us + uo];
        const float4 gv = *reinterpret_cast<const float4*>(L.gates + (size_t)t * 4 * us + (size_t)rr * N4 + unit * 4);       // gate-minor gates
        e[q].g[0] = gv.x; e[q].g[1] = gv.y; e[q].g[2] = gv.z; e[q].g[3] = gv.w;
        e[q].c = L.c[(size_t)t * us + uo];
        const float* cpp = t > 0 ? L.c + (size_t)(t - 1) * us + uo : (L.c0 != nullptr ? L.c0 + uo : zero);
        e[q].cp = *cpp;
        const uint8_t* mp = L.mask != nullptr ? L.mask + (size_t)t * us + uo : reinterpret_cast<const uint8_t*>(zero);
        e[q].keep = *mp;
    }
}

// KA = 4 U1 / 128, KB = 4 U2 / 128 (k-steps of 16 per wave, 8 waves).
template <int KA, int KB, typename F>
__global__ void __launch_bounds__(512) lstm2_persist_bwd_kernel(PBwdArgs A) {
    __shared__ BwdTiles S;
    const int nb1 = A.l1.U / 32, nb2 = A.l2.U / 32, nm = nb1 + nb2;
    const int grp = blockIdx.x % A.G, member = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B, nrt = A.nrt;
    const int Rv = (nrt - grp + A.G - 1) / A.G;
    const int n_items = T * Rv;
    unsigned* status = A.sync;
    unsigned* flags = A.sync + PST_FLAGS_OFF;
    const float* zero = reinterpret_cast<const float*>(A.sync + 4);
    if (threadIdx.x == 0) { S.abort = 0; S.local = 0; }
    __syncthreads();
    if (!pst_same_xcd(A.sync + A.xcc_off + grp * 32, status, member, nm, &S.abort, &S.local, A.sticky_off)) return;
    const bool local = A.allow_local && S.local != 0;
    float2* red2 = reinterpret_cast<float2*>(&S.red[0][0][0][0]);
    BwdTail tl;
    tl.valid = false;
    tl.dcv[0] = tl.dcv[1] = 0.f;
    tl.dbv[0] = tl.dbv[1] = tl.dbv[2] = tl.dbv[3] = 0.f;
    if (member < nb2) {
        // ---------------- layer 2 (leads) ----------------
        const PBwdLayer& L = A.l2;
        const int nt = member, U = L.U, N4 = 4 * U, unit = nt * 32 + r;
        const size_t slab = (size_t)(N4 / 16) * 1024;
        const int kb = w * 16 * KB + hh * 8;
        typename F::x8 bw[KB];
        load_frags_plain(L.wh_p + (size_t)unit * N4 + kb, bw);
        BwdEpi e[2], en[2];
        pb_epi_load<false>(L, B, nt, T - 1, grp * 32, zero, e);
        for (int i = 0; i < n_items; ++i) {
            const int k = i / Rv, t = T - 1 - k, rt = grp + A.G * (i - k * Rv), m0 = rt * 32;
            long long* trc = PST_TRP(2, member == 0 && rt == 0, k);
            PST_TR(trc, 0);
            if (k > 0 && !pst_wait(flags + rt * 32, status, nb2, (unsigned)k, nb2, 0u, &S.abort, A.sticky_off)) return;
            PST_TR(trc, 1);
            typename F::x8 a[KB];
            load_frags_xchg(k > 0 ? L.dzx + ((size_t)(t + 1) * nrt + rt) * slab : L.dzxT + (size_t)rt * slab, slab, w * KB, a);
            float dcl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ii = 2 * w + q;
                const size_t uo = (size_t)min(m0 + (ii & 3) + 8 * (ii >> 2) + 4 * hh, B - 1) * U + unit;
                const float* dp = (k == 0 || Rv == 1) ? zero : L.dc + uo;
                dcl[q] = *dp;
            }
            {
                const int i2 = min(i + 1, n_items - 1), k2 = i2 / Rv;
                pb_epi_load<false>(L, B, nt, T - 1 - k2, (grp + A.G * (i2 - k2 * Rv)) * 32, zero, en);
            }
            f32x16_t acc;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
            for (int s = 0; s < KB; ++s) acc = F::mfma32(a[s], bw[s], acc);
            pb_tail(L, S, B, Rv, nt, tl);                    // the previous item's plain stores issue while the MFMA chain runs
#pragma unroll
            for (int wc = 0; wc < 8; ++wc) red2[((0 * 8 + w) * 8 + wc) * 64 + lane] = make_float2(acc[2 * wc], acc[2 * wc + 1]);     // 8-byte slots: see the forward kernel
            lds_barrier();
            float dh[2], e_dc[2];
            float sum2[2] = {0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float2 p = red2[((0 * 8 + ww) * 8 + w) * 64 + lane];
                sum2[0] += p.x; sum2[1] += p.y;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // layer 2's mask, when given, says dh_ext is the gradient wrt the DROPPED output: dropout backward happens here (dy / kp * keep)
                dh[q] = (L.mask != nullptr ? e[q].dh / A.kp * (float)e[q].keep : e[q].dh) + sum2[q];
                e_dc[q] = (Rv == 1 && k > 0) ? tl.dcv[q] : dcl[q];
            }
            PST_TR(trc, 2);
            pb_finish<F>(L, S, B, nrt, t, rt, nt, dh, e, e_dc, tl, flags + rt * 32 + member, (unsigned)(k + 1), local, trc);
            e[0] = en[0]; e[1] = en[1];
            PST_TR(trc, 5);
        }
        pb_tail(L, S, B, Rv, nt, tl);
        pb_flush_db(L, S, nt, tl);
    } else {
        // ---------------- layer 1 ----------------
        const PBwdLayer& L = A.l1;
        const PBwdLayer& L2 = A.l2;
        const int nt = member - nb2, U = L.U, N4 = 4 * U, K2 = 4 * L2.U, unit = nt * 32 + r;
        const size_t slab = (size_t)(N4 / 16) * 1024, slab2 = (size_t)(K2 / 16) * 1024;
        const int kba = w * 16 * KA + hh * 8, kbb = w * 16 * KB + hh * 8;
        typename F::x8 bw[KA], bq[KB];
        load_frags_plain(L.wh_p + (size_t)unit * N4 + kba, bw);
        load_frags_plain(L2.wx_p + (size_t)unit * K2 + kbb, bq);
        for (int i = 0; i < n_items; ++i) {
            const int k = i / Rv, t = T - 1 - k, rt = grp + A.G * (i - k * Rv), m0 = rt * 32;
            long long* trc = PST_TRP(3, member == nb2 && rt == 0, k);
            PST_TR(trc, 0);
            if (!pst_wait(flags + rt * 32, status, nb2, (unsigned)(k + 1), nm, (unsigned)k, &S.abort, A.sticky_off)) return;
            PST_TR(trc, 1);
            typename F::x8 aq[KB], aw[KA];
            load_frags_xchg(L2.dzx + ((size_t)t * nrt + rt) * slab2, slab2, w * KB, aq);
            load_frags_xchg(k > 0 ? L.dzx + ((size_t)(t + 1) * nrt + rt) * slab : L.dzxT + (size_t)rt * slab, slab, w * KA, aw);
            float dcl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ii = 2 * w + q;
                const size_t uo = (size_t)min(m0 + (ii & 3) + 8 * (ii >> 2) + 4 * hh, B - 1) * U + unit;
                const float* dp = (k == 0 || Rv == 1) ? zero : L.dc + uo;
                dcl[q] = *dp;
            }
            BwdEpi e[2];
            pb_epi_load<true>(L, B, nt, t, m0, zero, e);      // same item, behind the hand-off loads: registers are scarce in this role
            f32x16_t accq, accw;
#pragma unroll
            for (int j = 0; j < 16; ++j) { accq[j] = 0.f; accw[j] = 0.f; }
#pragma unroll
            for (int s = 0; s < KB; ++s) accq = F::mfma32(aq[s], bq[s], accq);
#pragma unroll
            for (int s = 0; s < KA; ++s) accw = F::mfma32(aw[s], bw[s], accw);
            pb_tail(L, S, B, Rv, nt, tl);                    // the previous item's plain stores issue while the MFMA chains run
#pragma unroll
            for (int wc = 0; wc < 8; ++wc) {
                red2[((0 * 8 + w) * 8 + wc) * 64 + lane] = make_float2(accq[2 * wc], accq[2 * wc + 1]);
                red2[((1 * 8 + w) * 8 + wc) * 64 + lane] = make_float2(accw[2 * wc], accw[2 * wc + 1]);
            }
            lds_barrier();
            float dh[2], e_dc[2];
            float sq2[2] = {0.f, 0.f}, sw2[2] = {0.f, 0.f};
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) {
                const float2 pq = red2[((0 * 8 + ww) * 8 + w) * 64 + lane], pw = red2[((1 * 8 + ww) * 8 + w) * 64 + lane];
                sq2[0] += pq.x; sq2[1] += pq.y; sw2[0] += pw.x; sw2[1] += pw.y;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                dh[q] = (L.mask != nullptr ? sq2[q] * ((float)e[q].keep / A.kp) : sq2[q]) + sw2[q];
                e_dc[q] = (Rv == 1 && k > 0) ? tl.dcv[q] : dcl[q];
            }
            PST_TR(trc, 2);
            pb_finish<F>(L, S, B, nrt, t, rt, nt, dh, e, e_dc, tl, flags + rt * 32 + member, (unsigned)(k + 1), local, trc);
            PST_TR(trc, 5);
        }
        pb_tail(L, S, B, Rv, nt, tl);
        pb_flush_db(L, S, nt, tl);
    }
}

// Test hook (MNN_PERSIST_TEST_ABORT): raise the status word before the launch, so that every workgroup takes the give-up path.
__global__ void pst_poison_kernel(unsigned* sync) { sync[0] = 1u; }

// ---------------------------------------------------------------------------------------------- host side
static bool units_ok(int u) { return u == 128 || u == 256 || u == 512; }

static int cu_count() {             // of the CURRENT device
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

// G row-tile groups x (nb1 + nb2) members, R row tiles per workgroup; false when the grid cannot be resident at once.
static bool persist_plan(int B, int u1, int u2, int& nrt, int& G, int& R) {
    if (!units_ok(u1) || !units_ok(u2) || B <= 0) return false;
    const int nm = u1 / 32 + u2 / 32;
    if (nm > 32) return false;
    const int cus = cu_count();
    if (cus < nm) return false;
    nrt = cdiv(B, 32);
    const int gmax = cus / nm;
    R = cdiv(nrt, gmax);
    G = cdiv(nrt, R);
    return true;
}

// workspace: [status | 31 zero words][32 flags per row tile][boundary slabs: layer 1, layer 2 (zeros / h0)] -- one memset
// per call covers all of that -- [sticky word, padded][exchange area]
static size_t sync_words(int nrt) { return (size_t)PST_FLAGS_OFF + 64 * (size_t)nrt; }       // flags lines, then one XCC-id line per row-tile group
static size_t edge_bytes(int nrt, int u1, int u2, bool bwd) { return (size_t)nrt * (bwd ? 256 : 64) * ((size_t)u1 + u2); }
static size_t edge_max(int nrt, int u1, int u2) { return edge_bytes(nrt, u1, u2, true); }
static size_t sticky_offset(int nrt, int u1, int u2) { return sync_words(nrt) * sizeof(unsigned) + edge_max(nrt, u1, u2); }
static size_t xchg_offset(int nrt, int u1, int u2) { return (sticky_offset(nrt, u1, u2) + 16 + 255) / 256 * 256; }

extern "C" int mnn_lstm2_persist_ok(int B, int u1, int u2) {
    int nrt, G, R;
    return persist_plan(B, u1, u2, nrt, G, R) ? 1 : 0;
}

extern "C" size_t mnn_lstm2_persist_workspace_bytes(int T, int B, int u1, int u2) {
    const size_t nrt = (size_t)cdiv(B, 32), per = (size_t)T * nrt;
    const size_t fwd = per * 64 * (2 * (size_t)u1 + u2), bwd = per * 256 * ((size_t)u1 + u2);
    return xchg_offset((int)nrt, u1, u2) + (fwd > bwd ? fwd : bwd);
}

extern "C" int mnn_lstm2_persist_status(const void* workspace, int B, int u1, int u2, int* status) {
    MNN_REQUIRE(workspace && status && B > 0 && u1 > 0 && u2 > 0, "mnn_lstm2_persist_status: bad arguments");
    unsigned v = 0;
    MNN_HIP(hipMemcpy(&v, (const char*)workspace + sticky_offset(cdiv(B, 32), u1, u2), sizeof(v), hipMemcpyDeviceToHost));
    *status = (int)v;
    return MNN_OK;
}

template <int K1, typename F>
static hipError_t launch_pfwd(hipStream_t st, int grid, const PFwdArgs& a, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 8, F>), dim3(grid), dim3(256), 0, st, a);
    else if (u2 == 256) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 4, F>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 2, F>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
template <int KA, typename F>
static hipError_t launch_pbwd(hipStream_t st, int grid, const PBwdArgs& a, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 16, F>), dim3(grid), dim3(512), 0, st, a);
    else if (u2 == 256) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 8, F>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 4, F>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

static PFwdLayer fwd_layer(const mnn_lstm_fwd_layer* L) {
    PFwdLayer p{};
    p.xproj = L->xproj; p.wh_t = (const bf16_t*)L->wh_t; p.c0 = L->c0;
    p.gates = L->gates; p.c = L->c; p.h = (bf16_t*)L->h; p.hT = (bf16_t*)L->hT; p.ld_hT = L->ld_hT; p.y = (bf16_t*)L->y; p.mask = L->mask;
    p.wx_t = (const bf16_t*)L->wx_t; p.ld_w = L->ld_w; p.bias_p = L->bias_p; p.U = L->units;
    p.yT = (bf16_t*)L->yT; p.ld_yT = L->ld_yT;
    return p;
}

extern "C" int mnn_lstm2_persist_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L1, const mnn_lstm_fwd_layer* L2, float keep_prob,
                                     void* workspace) {
    MNN_REQUIRE(L1 && L2 && !L1->xproj_bf16 && !L2->xproj_bf16, "this form reads f32 input projections (xproj_bf16 is for mnn_lstm_rowpar_fwd)");
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_persist_fwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm2_persist_fwd: workspace must be 256-byte aligned");
    PFwdArgs a{};
    MNN_REQUIRE(persist_plan(B, L1->units, L2->units, a.nrt, a.G, a.R),
                "mnn_lstm2_persist_fwd: units must be 128/256/512 with (u1+u2)/32 <= 32 and the grid must fit the device (B=%d u=%d,%d)", B,
                L1->units, L2->units);
    MNN_REQUIRE(L1->xproj && L1->wh_t && L1->c && L1->h && L2->wh_t && L2->c && L2->h, "mnn_lstm2_persist_fwd: null pointer");
    MNN_REQUIRE(L2->wx_t && L2->bias_p && L2->ld_w >= L1->units, "mnn_lstm2_persist_fwd: layer 2 needs its input-projection weights");
    for (const mnn_lstm_fwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->hT == nullptr || L->ld_hT >= T * B, "mnn_lstm2_persist_fwd: ld_hT too small");
        MNN_REQUIRE(L->yT == nullptr || L->ld_yT >= T * B, "mnn_lstm2_persist_fwd: ld_yT too small");
        MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                    "mnn_lstm2_persist_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    }
    const int u1 = L1->units, u2 = L2->units;
    a.l1 = fwd_layer(L1); a.l2 = fwd_layer(L2);
    a.T = T; a.B = B; a.kp = keep_prob; a.sync = (unsigned*)workspace;
    a.xcc_off = PST_FLAGS_OFF + 32 * a.nrt; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    a.sticky_off = (int)(sticky_offset(a.nrt, L1->units, L2->units) / sizeof(unsigned));   // the aborting workgroup sets it; never re-zeroed
    const size_t per = (size_t)T * a.nrt;
    char* edge = (char*)workspace + sync_words(a.nrt) * sizeof(unsigned);
    char* x = (char*)workspace + xchg_offset(a.nrt, u1, u2);
    a.l1.hx0 = edge; a.l2.hx0 = edge + (size_t)a.nrt * 64 * u1;
    a.l1.hx = x; a.l1.yx = x + per * 64 * u1; a.l2.hx = x + per * 128 * u1; a.l2.yx = nullptr;    // layer 2's dropped output is not handed off
    const int grid = a.G * (u1 / 32 + u2 / 32);
    MNN_HIP(mnn_zero_async(workspace, sync_words(a.nrt) * sizeof(unsigned) + edge_bytes(a.nrt, u1, u2, false), st));
    if (getenv("MNN_PERSIST_TEST_ABORT")) hipLaunchKernelGGL(pst_poison_kernel, dim3(1), dim3(1), 0, st, (unsigned*)workspace);   // tests: exercise the give-up path
    if (L1->h0 || L2->h0)
        hipLaunchKernelGGL(pst_fill_h0_kernel, dim3(a.nrt, u1 / 16 + u2 / 16), dim3(64), 0, st, (const bf16_t*)L1->h0, u1, edge,
                           (const bf16_t*)L2->h0, u2, edge + (size_t)a.nrt * 64 * u1, B);
    MNN_REQUIRE((L1->f16 != 0) == (L2->f16 != 0), "mnn_lstm2_persist_fwd: both layers must use the same 16-bit flavour");
    hipError_t e;
    if (L1->f16) {
        if (u1 == 512) e = launch_pfwd<8, Fp16F>(st, grid, a, u2);
        else if (u1 == 256) e = launch_pfwd<4, Fp16F>(st, grid, a, u2);
        else e = launch_pfwd<2, Fp16F>(st, grid, a, u2);
    } else if (u1 == 512) e = launch_pfwd<8, Bf16F>(st, grid, a, u2);
    else if (u1 == 256) e = launch_pfwd<4, Bf16F>(st, grid, a, u2);
    else e = launch_pfwd<2, Bf16F>(st, grid, a, u2);
    MNN_HIP(e);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

static PBwdLayer bwd_layer(const mnn_lstm_bwd_layer* L) {
    PBwdLayer p{};
    p.dh_ext = L->dh_ext; p.wh_p = (const bf16_t*)L->wh_p; p.gates = L->gates; p.c = L->c; p.c0 = L->c0;
    p.dc = (float*)L->workspace; p.dzTt = (bf16_t*)L->dzT_t; p.ld_t = L->ld_t; p.mask = L->mask;
    p.wx_p = (const bf16_t*)L->wx_p; p.U = L->units; p.db_p = L->db_p;
    return p;
}

extern "C" int mnn_lstm2_persist_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L1, const mnn_lstm_bwd_layer* L2, float keep_prob,
                                     void* workspace) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && workspace && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_persist_bwd: bad arguments");
    MNN_REQUIRE(((size_t)workspace & 255) == 0, "mnn_lstm2_persist_bwd: workspace must be 256-byte aligned");
    PBwdArgs a{};
    MNN_REQUIRE(persist_plan(B, L1->units, L2->units, a.nrt, a.G, a.R),
                "mnn_lstm2_persist_bwd: units must be 128/256/512 with (u1+u2)/32 <= 32 and the grid must fit the device (B=%d u=%d,%d)", B,
                L1->units, L2->units);
    MNN_REQUIRE(L2->dh_ext && L2->wx_p, "mnn_lstm2_persist_bwd: layer 2 needs dh_ext and wx_p (its input weights, [u1, 4u2])");
    for (const mnn_lstm_bwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->wh_p && L->gates && L->c && L->workspace, "mnn_lstm2_persist_bwd: null pointer");
        MNN_REQUIRE(L->dz == nullptr, "mnn_lstm2_persist_bwd: an f32 dz output is not produced by the persistent form");
        MNN_REQUIRE(L->dzT_t == nullptr || L->ld_t >= T * B, "mnn_lstm2_persist_bwd: ld_t too small");
    }
    MNN_REQUIRE((L1->mask == nullptr) == (keep_prob >= 1.0f), "mnn_lstm2_persist_bwd: layer 1's keep mask is needed exactly when keep_prob < 1");
    const int u1 = L1->units, u2 = L2->units;
    a.l1 = bwd_layer(L1); a.l2 = bwd_layer(L2);
    a.T = T; a.B = B; a.kp = keep_prob; a.sync = (unsigned*)workspace;
    a.xcc_off = PST_FLAGS_OFF + 32 * a.nrt; a.allow_local = getenv("MNN_PERSIST_NO_LOCAL") == nullptr;
    a.sticky_off = (int)(sticky_offset(a.nrt, L1->units, L2->units) / sizeof(unsigned));   // the aborting workgroup sets it; never re-zeroed
    const size_t per = (size_t)T * a.nrt;
    char* edge = (char*)workspace + sync_words(a.nrt) * sizeof(unsigned);
    char* x = (char*)workspace + xchg_offset(a.nrt, u1, u2);
    a.l1.dzxT = edge; a.l2.dzxT = edge + (size_t)a.nrt * 256 * u1;
    a.l1.dzx = x; a.l2.dzx = x + per * 256 * u1;
    const int grid = a.G * (u1 / 32 + u2 / 32);
    MNN_HIP(mnn_zero_async(workspace, sync_words(a.nrt) * sizeof(unsigned) + edge_bytes(a.nrt, u1, u2, true), st));
    MNN_REQUIRE((L1->f16 != 0) == (L2->f16 != 0), "mnn_lstm2_persist_bwd: both layers must use the same 16-bit flavour");
    hipError_t e;
    if (L1->f16) {
        if (u1 == 512) e = launch_pbwd<16, Fp16F>(st, grid, a, u2);
        else if (u1 == 256) e = launch_pbwd<8, Fp16F>(st, grid, a, u2);
        else e = launch_pbwd<4, Fp16F>(st, grid, a, u2);
    } else if (u1 == 512) e = launch_pbwd<16, Bf16F>(st, grid, a, u2);
    else if (u1 == 256) e = launch_pbwd<8, Bf16F>(st, grid, a, u2);
    else e = launch_pbwd<4, Bf16F>(st, grid, a, u2);
    MNN_HIP(e);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
