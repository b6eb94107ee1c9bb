// Persistent two-layer LSTM recurrence for gfx950 (bf16 operands, f32 state).
//
// The per-timestep launches of gemm.hip re-read every weight from L2/MALL at each step and pay a kernel boundary per
// step; the recurrence is a latency chain, so both sit on the critical path.  Here ONE launch runs all T steps:
//   * a workgroup owns one (unit tile, row tile) of one layer for the whole sequence and keeps its slice of the
//     recurrent weights in REGISTERS (32 units x K, K split over the waves) -- weights are read once per launch;
//   * the 32-row hidden-state tiles move between workgroups through global memory: 16-byte write-through (sc1)
//     stores, `s_waitcnt vmcnt(0)` in every storing wave, workgroup barrier, one sc1 flag store per workgroup;
//     consumers poll the flags of their row tile with sc1 loads from one wave, join a barrier, then read the tile with
//     16-byte sc1 buffer loads (MI355X_MICROARCH.md, "Valid forms", first table row; cdna_hip_programming.md G16 R1);
//   * only workgroups of the same ROW TILE ever wait for each other (16 + 8 of them for units 512/256), never the grid;
//   * layer 2 consumes layer 1's step t as soon as its flags say so: its input projection is folded into its step
//     (K = U1 + U2), so the xproj round trip of the launch-per-step form disappears too.
// Every spin is bounded (1 s of the 100 MHz realtime counter) and watches a status word: a workgroup that gives up
// sets it and every other one leaves at its next poll, so the grid always drains.  The host entry refuses shapes whose
// grid is not resident at once (one workgroup per CU).  Block -> tile map: blocks with equal (id % G) share a row tile,
// so with G = 8 a row tile's workgroups share an XCD under round-robin dispatch (speed only, never correctness).
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32;

#define PST_LIMIT 100000000LL      // spin bound: 1 s of wall_clock64() (100 MHz)
#define PST_FLAGS_OFF 32           // words: [0] status, [32 + 32*rt + member] progress flags, then one sticky word
#define PST_SC1 16                 // aux bit of raw buffer loads/stores: device-scope (write-through / L1-bypassing)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load((gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) { __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t slice_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
template <int KS>
__device__ __forceinline__ void load_frags_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, bf16x8_t (&f)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off + 32 * s, 0, PST_SC1));
}
template <int KS>
__device__ __forceinline__ void load_frags_plain(const bf16_t* __restrict__ p, bf16x8_t (&f)[KS]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *reinterpret_cast<const bf16x8_t*>(p + 16 * s);
}

// Wave 0 polls one 128-byte line of progress flags: lanes [0,n1) need >= need1, lanes [n1,nm) need >= need2, the rest
// watch the status word.  All threads of the workgroup call this; returns false (uniformly) when the launch is aborting.
__device__ __forceinline__ bool pst_wait(const unsigned* line, unsigned* status, int n1, unsigned need1, int nm, unsigned need2, int* s_abort) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool mine = lane < nm;
        const unsigned need = lane < n1 ? need1 : need2;
        const unsigned* p = mine ? line + lane : status;
        const long long t0 = wall_clock64();
        for (unsigned spins = 1;; ++spins) {
            const unsigned v = ld_agent(p);
            if (__all(mine ? v >= need : v == 0u)) break;
            const bool dead = __any(!mine && v != 0u) || ((spins & 127u) == 0u && wall_clock64() - t0 > PST_LIMIT);
            if (dead) {
                if (lane == 0) { st_agent(status, 1u); *s_abort = 1; }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return *reinterpret_cast<volatile int*>(s_abort) == 0;
}

// Publish: every storing wave drains its write-through stores, the workgroup meets, one lane raises the flag.
__device__ __forceinline__ void pst_publish(unsigned* flag, unsigned value) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_agent(flag, value);
}

struct PFwdLayer {
    const float* xproj; const bf16_t* wh_t; const bf16_t* h0; const float* c0;
    float* gates; float* c; bf16_t* h; bf16_t* hT; int ld_hT; bf16_t* y; const uint8_t* mask;
    const bf16_t* wx_t; int ld_w; const float* bias_p; int U;
};
struct PFwdArgs { PFwdLayer l1, l2; int T, B, nrt, G, R; float kp; unsigned* sync; };

struct FwdTiles {
    float red[4][4][16][64];       // K-split partial tiles
    bf16_t sH[32][40];             // h tile [row][unit] for the 16-byte write-through stores
    bf16_t sY[32][40];             // dropped output tile
    bf16_t sT[32][40];             // h tile [unit][row] for the transposed copy (weight-gradient operand)
    int abort;
};

// Gate pointwise + the stores of one (t, row tile) for one workgroup; z = pre-activations of this wave's 4 fragment rows.
__device__ __forceinline__ void pf_finish(const PFwdLayer& L, FwdTiles& S, int T, int B, int t, int m0, int nt, float kp, const float (&z)[4][4],
                                          const float (&cp)[4], const float (&mk)[4], const bool (&live)[4], unsigned* flag) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int U = L.U, N4 = 4 * U, n0 = nt * 128, unit = nt * 32 + r;
    const size_t us = (size_t)B * U;
    float gv[4][4], cv[4];
    const bool drop = L.mask != nullptr;
    const bool wantT = L.hT != nullptr && t + 1 < T;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float gi = fast_sigmoid(z[q][0]), gg = fast_tanh(z[q][1]), gf = fast_sigmoid(z[q][2]), go = fast_sigmoid(z[q][3]);
        const float c = gg * gi + cp[q] * gf;
        const float h = fast_tanh(c) * go;
        gv[q][0] = gi; gv[q][1] = gg; gv[q][2] = gf; gv[q][3] = go; cv[q] = c;
        const bf16_t hb = f32_to_bf16(h);
        const int lr = 8 * w + q + 4 * hh;
        S.sH[lr][r] = hb;
        if (drop) S.sY[lr][r] = f32_to_bf16(bf16_to_f32(hb) / kp * mk[q]);
        if (wantT) S.sT[r][lr] = hb;
    }
    __syncthreads();
    {   // handed-off tiles: 32 rows x 64 bytes each, one 16-byte write-through store per thread (h: threads 0..127, y: 128..255)
        const int tt = threadIdx.x & 127, row = tt >> 2, piece = tt & 3;
        const bool second = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 7)) != 0;      // wave-uniform: keeps the resource in SGPRs
        if (m0 + row < B && (!second || drop)) {
            bf16_t* base = (second ? L.y : L.h) + (size_t)t * us;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(second ? &S.sY[row][piece * 8] : &S.sH[row][piece * 8]);
            __builtin_amdgcn_raw_buffer_store_b128(v, slice_rsrc(base, us * 2), (int)(((size_t)(m0 + row) * U + nt * 32 + piece * 8) * 2), 0, PST_SC1);
        }
    }
    pst_publish(flag, (unsigned)(t + 1));
    // everything below is consumed after the launch (or by this workgroup only): plain stores, off the critical chain
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!live[q]) continue;
        const int row = m0 + 8 * w + q + 4 * hh;
        const size_t zo = (size_t)t * 4 * us + (size_t)row * N4 + n0 + r, uo = (size_t)t * us + (size_t)row * U + unit;
        if (L.gates != nullptr) { L.gates[zo] = gv[q][0]; L.gates[zo + 32] = gv[q][1]; L.gates[zo + 64] = gv[q][2]; L.gates[zo + 96] = gv[q][3]; }
        L.c[uo] = cv[q];
    }
    if (wantT && threadIdx.x < 128) {      // hT[unit][(t+1) B + row]
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

// KS1 = U1 / 64, KS2 = U2 / 64 (k-steps of 16 per wave, 4 waves).
template <int KS1, int KS2>
__global__ void __launch_bounds__(256) lstm2_persist_fwd_kernel(PFwdArgs A) {
    __shared__ FwdTiles S;
    const int nb1 = A.l1.U / 32, nb2 = A.l2.U / 32, nm = nb1 + nb2;
    const int grp = blockIdx.x % A.G, member = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B;
    unsigned* status = A.sync;
    unsigned* flags = A.sync + PST_FLAGS_OFF;
    if (threadIdx.x == 0) S.abort = 0;
    __syncthreads();
    if (member < nb1) {
        // ---------------- layer 1: z = xproj[t] + h[t-1] . Wh^T ----------------
        const PFwdLayer& L = A.l1;
        const int nt = member, U = L.U, N4 = 4 * U, n0 = nt * 128, unit = nt * 32 + r;
        const int kb = w * 16 * KS1 + hh * 8;
        const size_t us = (size_t)B * U;
        bf16x8_t b[4][KS1];
#pragma unroll
        for (int g = 0; g < 4; ++g) load_frags_plain<KS1>(L.wh_t + (size_t)(n0 + 32 * g + r) * U + kb, b[g]);
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < A.R; ++j) {
                const int rt = grp + A.G * j;
                if (rt >= A.nrt) break;
                const int m0 = rt * 32;
                float xp[4][4], cp[4], mk[4];
                bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + 8 * w + q + 4 * hh;
                    live[q] = row < B;
                    const int rr = live[q] ? row : B - 1;
                    const size_t zo = (size_t)t * 4 * us + (size_t)rr * N4 + n0 + r, uo = (size_t)rr * U + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) xp[q][g] = L.xproj[zo + 32 * g];
                    cp[q] = t > 0 ? L.c[(size_t)(t - 1) * us + uo] : (L.c0 != nullptr ? L.c0[uo] : 0.f);
                    mk[q] = L.mask != nullptr ? (float)L.mask[(size_t)t * us + uo] : 1.f;
                }
                f32x16_t acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
                if (t > 0 && !pst_wait(flags + rt * 32, status, nb1, (unsigned)t, nb1, 0u, &S.abort)) return;
                const bf16_t* hp = t > 0 ? L.h + (size_t)(t - 1) * us : L.h0;
                if (hp != nullptr) {
                    const int arow = min(m0 + r, B - 1);
                    bf16x8_t a[KS1];
                    load_frags_sc1<KS1>(slice_rsrc(hp, us * 2), (arow * U + kb) * 2, a);
#pragma unroll
                    for (int s = 0; s < KS1; ++s)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[g][s], acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) S.red[w][g][i][lane] = acc[g][i];
                __syncthreads();
                float z[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * w + q;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        z[q][g] = xp[q][g] + ((S.red[0][g][i][lane] + S.red[1][g][i][lane]) + (S.red[2][g][i][lane] + S.red[3][g][i][lane]));
                }
                pf_finish(L, S, T, B, t, m0, nt, A.kp, z, cp, mk, live, flags + rt * 32 + member);
            }
        }
    } else {
        // ---------------- layer 2: z = bias + y1[t] . Wx^T + h[t-1] . Wh^T ----------------
        const PFwdLayer& L = A.l2;
        const PFwdLayer& L1 = A.l1;
        const int nt = member - nb1, U = L.U, U1 = L1.U, n0 = nt * 128, unit = nt * 32 + r;
        const int kb1 = w * 16 * KS1 + hh * 8, kb2 = w * 16 * KS2 + hh * 8;
        const size_t us = (size_t)B * U, us1 = (size_t)B * U1;
        const bf16_t* y1 = L1.mask != nullptr ? L1.y : L1.h;
        bf16x8_t bx[4][KS1], bh[4][KS2];
        float bz[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            load_frags_plain<KS1>(L.wx_t + (size_t)(n0 + 32 * g + r) * L.ld_w + kb1, bx[g]);
            load_frags_plain<KS2>(L.wh_t + (size_t)(n0 + 32 * g + r) * U + kb2, bh[g]);
            bz[g] = L.bias_p[n0 + 32 * g + r];
        }
        for (int t = 0; t < T; ++t) {
            for (int j = 0; j < A.R; ++j) {
                const int rt = grp + A.G * j;
                if (rt >= A.nrt) break;
                const int m0 = rt * 32;
                float cp[4], mk[4];
                bool live[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + 8 * w + q + 4 * hh;
                    live[q] = row < B;
                    const int rr = live[q] ? row : B - 1;
                    const size_t uo = (size_t)rr * U + unit;
                    cp[q] = t > 0 ? L.c[(size_t)(t - 1) * us + uo] : (L.c0 != nullptr ? L.c0[uo] : 0.f);
                    mk[q] = L.mask != nullptr ? (float)L.mask[(size_t)t * us + uo] : 1.f;
                }
                f32x16_t acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;
                if (!pst_wait(flags + rt * 32, status, nb1, (unsigned)(t + 1), nm, (unsigned)t, &S.abort)) return;
                const int arow = min(m0 + r, B - 1);
                const bf16_t* hp = t > 0 ? L.h + (size_t)(t - 1) * us : L.h0;
                bf16x8_t a1[KS1], a2[KS2];
                load_frags_sc1<KS1>(slice_rsrc(y1 + (size_t)t * us1, us1 * 2), (arow * U1 + kb1) * 2, a1);
                if (hp != nullptr) load_frags_sc1<KS2>(slice_rsrc(hp, us * 2), (arow * U + kb2) * 2, a2);
#pragma unroll
                for (int s = 0; s < KS1; ++s)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], bx[g][s], acc[g], 0, 0, 0);
                if (hp != nullptr) {
#pragma unroll
                    for (int s = 0; s < KS2; ++s)
#pragma unroll
                        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s], bh[g][s], acc[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) S.red[w][g][i][lane] = acc[g][i];
                __syncthreads();
                float z[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * w + q;
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        z[q][g] = bz[g] + ((S.red[0][g][i][lane] + S.red[1][g][i][lane]) + (S.red[2][g][i][lane] + S.red[3][g][i][lane]));
                }
                pf_finish(L, S, T, B, t, m0, nt, A.kp, z, cp, mk, live, flags + rt * 32 + member);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward.  512-thread workgroups (K split over 8 waves), one 32-row x 32-unit tile of dh each:
//   layer 2, step t:  dh = dh_ext[t] + dz2[t+1] . Wh2            (K = 4 U2)
//   layer 1, step t:  dh = (dz2[t] . Wx2) * keep/kp + dz1[t+1] . Wh1   (K = 4 U2 + 4 U1; the dgrad of layer 2's input
//                     projection through the dropout of rnn.py:132 is folded in)
// followed by the gate pointwise; the handed-off tile is dz (bf16, [B, 4U], gate-interleaved: 256 bytes per row per tile).
// ------------------------------------------------------------------------------------------------------------------
struct PBwdLayer {
    const float* dh_ext; const bf16_t* wh_p; const float* gates; const float* c; const float* c0;
    float* dc; float* dz; bf16_t* dzc; bf16_t* dzTt; int ld_t; const uint8_t* mask; const bf16_t* wx_p; int U;
};
struct PBwdArgs { PBwdLayer l1, l2; int T, B, nrt, G, R; float kp; unsigned* sync; };

struct BwdTiles {
    float red[2][8][16][64];
    bf16_t sZ[32][136];            // dz tile [row][gate*32 + unit] (+pad) for the 16-byte write-through stores
    bf16_t sT[4][32][40];          // dz tile [gate][unit][row] for the transposed copy
    int abort;
};

__device__ __forceinline__ void pb_finish(const PBwdLayer& L, BwdTiles& S, int T, int B, int t, int m0, int nt, const float (&dh)[2],
                                          const float (&e_g)[2][4], const float (&e_c)[2], const float (&e_cp)[2], const float (&e_dc)[2],
                                          const bool (&live)[2], unsigned* flag, unsigned epoch) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int U = L.U, N4 = 4 * U, unit = nt * 32 + r, pc = nt * 128 + r;
    const size_t us = (size_t)B * U;
    float dzv[2][4], dcv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = 2 * w + q;
        const int lr = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float gi = e_g[q][0], gg = e_g[q][1], gf = e_g[q][2], go = e_g[q][3];
        const float tc = fast_tanh(e_c[q]);
        const float d_o = dh[q] * tc;
        const float d_c = dh[q] * go * (1.f - tc * tc) + e_dc[q];
        dzv[q][0] = d_c * gg * gi * (1.f - gi);
        dzv[q][1] = d_c * gi * (1.f - gg * gg);
        dzv[q][2] = d_c * e_cp[q] * gf * (1.f - gf);
        dzv[q][3] = d_o * go * (1.f - go);
        dcv[q] = d_c * gf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bf16_t bv = f32_to_bf16(dzv[q][g]);
            S.sZ[lr][32 * g + r] = bv;
            if (L.dzTt != nullptr) S.sT[g][r][lr] = bv;
        }
    }
    __syncthreads();
    {   // 32 rows x 256 bytes: one 16-byte write-through store per thread
        const int row = threadIdx.x >> 4, piece = threadIdx.x & 15;
        if (m0 + row < B) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&S.sZ[row][piece * 8]);
            __builtin_amdgcn_raw_buffer_store_b128(v, slice_rsrc(L.dzc + (size_t)t * 4 * us, 4 * us * 2),
                                                   (int)(((size_t)(m0 + row) * N4 + nt * 128 + piece * 8) * 2), 0, PST_SC1);
        }
    }
    pst_publish(flag, epoch);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!live[q]) continue;
        const int i = 2 * w + q;
        const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
        const size_t uo = (size_t)row * U + unit, zo = (size_t)t * 4 * us + (size_t)row * N4 + pc;
        L.dc[uo] = dcv[q];
        if (L.dz != nullptr) { L.dz[zo] = dzv[q][0]; L.dz[zo + 32] = dzv[q][1]; L.dz[zo + 64] = dzv[q][2]; L.dz[zo + 96] = dzv[q][3]; }
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

// KA = 4 U1 / 128, KB = 4 U2 / 128 (k-steps of 16 per wave, 8 waves).
template <int KA, int KB>
__global__ void __launch_bounds__(512) lstm2_persist_bwd_kernel(PBwdArgs A) {
    __shared__ BwdTiles S;
    const int nb1 = A.l1.U / 32, nb2 = A.l2.U / 32, nm = nb1 + nb2;
    const int grp = blockIdx.x % A.G, member = blockIdx.x / A.G;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int T = A.T, B = A.B;
    unsigned* status = A.sync;
    unsigned* flags = A.sync + PST_FLAGS_OFF;
    if (threadIdx.x == 0) S.abort = 0;
    __syncthreads();
    if (member < nb2) {
        // ---------------- layer 2 ----------------
        const PBwdLayer& L = A.l2;
        const int nt = member, U = L.U, N4 = 4 * U, unit = nt * 32 + r, pc = nt * 128 + r;
        const int kb = w * 16 * KB + hh * 8;
        const size_t us = (size_t)B * U;
        bf16x8_t bw[KB];
        load_frags_plain<KB>(L.wh_p + (size_t)unit * N4 + kb, bw);
        for (int k = 0; k < T; ++k) {
            const int t = T - 1 - k;
            for (int j = 0; j < A.R; ++j) {
                const int rt = grp + A.G * j;
                if (rt >= A.nrt) break;
                const int m0 = rt * 32;
                float e_dh[2], e_g[2][4], e_c[2], e_cp[2], e_dc[2];
                bool live[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = 2 * w + q;
                    const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    live[q] = row < B;
                    const int rr = live[q] ? row : B - 1;
                    const size_t uo = (size_t)rr * U + unit, zo = (size_t)t * 4 * us + (size_t)rr * N4 + pc;
                    e_dh[q] = L.dh_ext[(size_t)t * us + uo];
#pragma unroll
                    for (int g = 0; g < 4; ++g) e_g[q][g] = L.gates[zo + 32 * g];
                    e_c[q] = L.c[(size_t)t * us + uo];
                    e_cp[q] = t > 0 ? L.c[(size_t)(t - 1) * us + uo] : (L.c0 != nullptr ? L.c0[uo] : 0.f);
                    e_dc[q] = k > 0 ? L.dc[uo] : 0.f;
                }
                f32x16_t acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                if (k > 0) {
                    if (!pst_wait(flags + rt * 32, status, nb2, (unsigned)k, nb2, 0u, &S.abort)) return;
                    const int arow = min(m0 + r, B - 1);
                    bf16x8_t a[KB];
                    load_frags_sc1<KB>(slice_rsrc(L.dzc + (size_t)(t + 1) * 4 * us, 4 * us * 2), (arow * N4 + kb) * 2, a);
#pragma unroll
                    for (int s = 0; s < KB; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bw[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) S.red[0][w][i][lane] = acc[i];
                __syncthreads();
                float dh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = 2 * w + q;
                    float sum = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) sum += S.red[0][ww][i][lane];
                    dh[q] = e_dh[q] + sum;
                }
                pb_finish(L, S, T, B, t, m0, nt, dh, e_g, e_c, e_cp, e_dc, live, flags + rt * 32 + member, (unsigned)(k + 1));
            }
        }
    } else {
        // ---------------- layer 1 ----------------
        const PBwdLayer& L = A.l1;
        const PBwdLayer& L2 = A.l2;
        const int nt = member - nb2, U = L.U, N4 = 4 * U, K2 = 4 * L2.U, unit = nt * 32 + r, pc = nt * 128 + r;
        const int kba = w * 16 * KA + hh * 8, kbb = w * 16 * KB + hh * 8;
        const size_t us = (size_t)B * U, us2 = (size_t)B * L2.U;
        bf16x8_t bw[KA], bq[KB];
        load_frags_plain<KA>(L.wh_p + (size_t)unit * N4 + kba, bw);
        load_frags_plain<KB>(L2.wx_p + (size_t)unit * K2 + kbb, bq);
        for (int k = 0; k < T; ++k) {
            const int t = T - 1 - k;
            for (int j = 0; j < A.R; ++j) {
                const int rt = grp + A.G * j;
                if (rt >= A.nrt) break;
                const int m0 = rt * 32;
                float e_g[2][4], e_c[2], e_cp[2], e_dc[2], e_f[2];
                bool live[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = 2 * w + q;
                    const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    live[q] = row < B;
                    const int rr = live[q] ? row : B - 1;
                    const size_t uo = (size_t)rr * U + unit, zo = (size_t)t * 4 * us + (size_t)rr * N4 + pc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) e_g[q][g] = L.gates[zo + 32 * g];
                    e_c[q] = L.c[(size_t)t * us + uo];
                    e_cp[q] = t > 0 ? L.c[(size_t)(t - 1) * us + uo] : (L.c0 != nullptr ? L.c0[uo] : 0.f);
                    e_dc[q] = k > 0 ? L.dc[uo] : 0.f;
                    e_f[q] = L.mask != nullptr ? (float)L.mask[(size_t)t * us + uo] / A.kp : 1.f;
                }
                f32x16_t accq, accw;
#pragma unroll
                for (int i = 0; i < 16; ++i) { accq[i] = 0.f; accw[i] = 0.f; }
                if (!pst_wait(flags + rt * 32, status, nb2, (unsigned)(k + 1), nm, (unsigned)k, &S.abort)) return;
                const int arow = min(m0 + r, B - 1);
                {
                    bf16x8_t aq[KB];
                    load_frags_sc1<KB>(slice_rsrc(L2.dzc + (size_t)t * 4 * us2, 4 * us2 * 2), (arow * K2 + kbb) * 2, aq);
                    if (k > 0) {
                        bf16x8_t aw[KA];
                        load_frags_sc1<KA>(slice_rsrc(L.dzc + (size_t)(t + 1) * 4 * us, 4 * us * 2), (arow * N4 + kba) * 2, aw);
#pragma unroll
                        for (int s = 0; s < KB; ++s) accq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[s], bq[s], accq, 0, 0, 0);
#pragma unroll
                        for (int s = 0; s < KA; ++s) accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[s], bw[s], accw, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int s = 0; s < KB; ++s) accq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[s], bq[s], accq, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { S.red[0][w][i][lane] = accq[i]; S.red[1][w][i][lane] = accw[i]; }
                __syncthreads();
                float dh[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = 2 * w + q;
                    float sq = 0.f, sw = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) { sq += S.red[0][ww][i][lane]; sw += S.red[1][ww][i][lane]; }
                    dh[q] = (L.mask != nullptr ? sq * e_f[q] : sq) + sw;
                }
                pb_finish(L, S, T, B, t, m0, nt, dh, e_g, e_c, e_cp, e_dc, live, flags + rt * 32 + member, (unsigned)(k + 1));
            }
        }
    }
}

// db_p[c] += sum over columns [c0, c1) of row c of dz^T (the LSTM bias gradient; same pass as gemm.hip's step-per-launch form)
__global__ void __launch_bounds__(256) pst_rowsum_bf16_kernel(const bf16_t* __restrict__ X, int ld, int c0, int c1, float* __restrict__ out) {
    __shared__ float part[4];
    const bf16_t* row = X + (size_t)blockIdx.x * ld;
    float acc = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) acc += bf16_to_f32(row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] += (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ void pst_sticky_kernel(unsigned* sync, int sticky_off) {
    if (sync[0] != 0u) sync[sticky_off] = sync[0];
}

// ---------------------------------------------------------------------------------------------- host side
static bool units_ok(int u) { return u == 128 || u == 256 || u == 512; }

static int cu_count() {
    static int n = -1;
    if (n < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        n = p.multiProcessorCount;
    }
    return n;
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

extern "C" int mnn_lstm2_persist_ok(int B, int u1, int u2) {
    int nrt, G, R;
    return persist_plan(B, u1, u2, nrt, G, R) ? 1 : 0;
}

extern "C" size_t mnn_lstm2_persist_sync_bytes(int B) {
    return ((size_t)PST_FLAGS_OFF + 32 * (size_t)cdiv(B, 32) + 4) * sizeof(unsigned);
}

template <int K1>
static hipError_t launch_pfwd(hipStream_t st, int grid, const PFwdArgs& a, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 8>), dim3(grid), dim3(256), 0, st, a);
    else if (u2 == 256) hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((lstm2_persist_fwd_kernel<K1, 2>), dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}
template <int KA>
static hipError_t launch_pbwd(hipStream_t st, int grid, const PBwdArgs& a, int u2) {
    if (u2 == 512) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 16>), dim3(grid), dim3(512), 0, st, a);
    else if (u2 == 256) hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 8>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((lstm2_persist_bwd_kernel<KA, 4>), dim3(grid), dim3(512), 0, st, a);
    return hipGetLastError();
}

static PFwdLayer fwd_layer(const mnn_lstm_fwd_layer* L) {
    PFwdLayer p{};
    p.xproj = L->xproj; p.wh_t = (const bf16_t*)L->wh_t; p.h0 = (const bf16_t*)L->h0; p.c0 = L->c0;
    p.gates = L->gates; p.c = L->c; p.h = (bf16_t*)L->h; p.hT = (bf16_t*)L->hT; p.ld_hT = L->ld_hT; p.y = (bf16_t*)L->y; p.mask = L->mask;
    p.wx_t = (const bf16_t*)L->wx_t; p.ld_w = L->ld_w; p.bias_p = L->bias_p; p.U = L->units;
    return p;
}

extern "C" int mnn_lstm2_persist_fwd(mnn_stream_t s, int T, int B, const mnn_lstm_fwd_layer* L1, const mnn_lstm_fwd_layer* L2, float keep_prob,
                                     void* sync) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && sync && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_persist_fwd: bad arguments");
    PFwdArgs a{};
    MNN_REQUIRE(persist_plan(B, L1->units, L2->units, a.nrt, a.G, a.R),
                "mnn_lstm2_persist_fwd: units must be 128/256/512 with (u1+u2)/32 <= 32 and the grid must fit the device (B=%d u=%d,%d)", B,
                L1->units, L2->units);
    MNN_REQUIRE(L1->xproj && L1->wh_t && L1->c && L1->h && L2->wh_t && L2->c && L2->h, "mnn_lstm2_persist_fwd: null pointer");
    MNN_REQUIRE(L2->wx_t && L2->bias_p && L2->ld_w >= L1->units, "mnn_lstm2_persist_fwd: layer 2 needs its input-projection weights");
    MNN_REQUIRE((size_t)T * B * 4 * (L1->units > L2->units ? L1->units : L2->units) * 4 < ((size_t)1 << 40), "mnn_lstm2_persist_fwd: sequence too large");
    MNN_REQUIRE((size_t)B * L1->units * 2 < ((size_t)1 << 31), "mnn_lstm2_persist_fwd: B*units too large for 32-bit tile offsets");
    for (const mnn_lstm_fwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->hT == nullptr || L->ld_hT >= T * B, "mnn_lstm2_persist_fwd: ld_hT too small");
        MNN_REQUIRE((L->mask == nullptr) == (keep_prob >= 1.0f) && (L->mask == nullptr || L->y != nullptr),
                    "mnn_lstm2_persist_fwd: a keep mask and a y buffer are needed exactly when keep_prob < 1");
    }
    a.l1 = fwd_layer(L1); a.l2 = fwd_layer(L2);
    a.T = T; a.B = B; a.kp = keep_prob; a.sync = (unsigned*)sync;
    const int grid = a.G * (L1->units / 32 + L2->units / 32);
    const size_t zero_bytes = ((size_t)PST_FLAGS_OFF + 32 * (size_t)a.nrt) * sizeof(unsigned);
    MNN_HIP(hipMemsetAsync(sync, 0, zero_bytes, st));
    hipError_t e;
    if (L1->units == 512) e = launch_pfwd<8>(st, grid, a, L2->units);
    else if (L1->units == 256) e = launch_pfwd<4>(st, grid, a, L2->units);
    else e = launch_pfwd<2>(st, grid, a, L2->units);
    MNN_HIP(e);
    hipLaunchKernelGGL(pst_sticky_kernel, dim3(1), dim3(1), 0, st, (unsigned*)sync, PST_FLAGS_OFF + 32 * a.nrt);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

static PBwdLayer bwd_layer(const mnn_lstm_bwd_layer* L) {
    PBwdLayer p{};
    p.dh_ext = L->dh_ext; p.wh_p = (const bf16_t*)L->wh_p; p.gates = L->gates; p.c = L->c; p.c0 = L->c0;
    p.dc = (float*)L->workspace; p.dz = L->dz; p.dzc = (bf16_t*)L->dz_T; p.dzTt = (bf16_t*)L->dzT_t; p.ld_t = L->ld_t; p.mask = L->mask;
    p.wx_p = (const bf16_t*)L->wx_p; p.U = L->units;
    return p;
}

extern "C" int mnn_lstm2_persist_bwd(mnn_stream_t s, int T, int B, const mnn_lstm_bwd_layer* L1, const mnn_lstm_bwd_layer* L2, float keep_prob,
                                     void* sync) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(L1 && L2 && sync && T > 0 && B > 0 && keep_prob > 0.f, "mnn_lstm2_persist_bwd: bad arguments");
    PBwdArgs a{};
    MNN_REQUIRE(persist_plan(B, L1->units, L2->units, a.nrt, a.G, a.R),
                "mnn_lstm2_persist_bwd: units must be 128/256/512 with (u1+u2)/32 <= 32 and the grid must fit the device (B=%d u=%d,%d)", B,
                L1->units, L2->units);
    MNN_REQUIRE(L2->dh_ext && L2->wx_p, "mnn_lstm2_persist_bwd: layer 2 needs dh_ext and wx_p (its input weights, [u1, 4u2])");
    MNN_REQUIRE((size_t)B * 4 * L1->units * 2 < ((size_t)1 << 31), "mnn_lstm2_persist_bwd: B*units too large for 32-bit tile offsets");
    for (const mnn_lstm_bwd_layer* L : {L1, L2}) {
        MNN_REQUIRE(L->wh_p && L->gates && L->c && L->dz_T && L->workspace, "mnn_lstm2_persist_bwd: null pointer");
        MNN_REQUIRE(L->dzT_t == nullptr || L->ld_t >= T * B, "mnn_lstm2_persist_bwd: ld_t too small");
        MNN_REQUIRE(L->db_p == nullptr || L->dzT_t != nullptr, "mnn_lstm2_persist_bwd: db_p needs dzT_t");
    }
    MNN_REQUIRE((L1->mask == nullptr) == (keep_prob >= 1.0f), "mnn_lstm2_persist_bwd: layer 1's keep mask is needed exactly when keep_prob < 1");
    a.l1 = bwd_layer(L1); a.l2 = bwd_layer(L2);
    a.T = T; a.B = B; a.kp = keep_prob; a.sync = (unsigned*)sync;
    const int grid = a.G * (L1->units / 32 + L2->units / 32);
    const size_t zero_bytes = ((size_t)PST_FLAGS_OFF + 32 * (size_t)a.nrt) * sizeof(unsigned);
    MNN_HIP(hipMemsetAsync(sync, 0, zero_bytes, st));
    hipError_t e;
    if (L1->units == 512) e = launch_pbwd<16>(st, grid, a, L2->units);
    else if (L1->units == 256) e = launch_pbwd<8>(st, grid, a, L2->units);
    else e = launch_pbwd<4>(st, grid, a, L2->units);
    MNN_HIP(e);
    hipLaunchKernelGGL(pst_sticky_kernel, dim3(1), dim3(1), 0, st, (unsigned*)sync, PST_FLAGS_OFF + 32 * a.nrt);
    if (L2->db_p) hipLaunchKernelGGL(pst_rowsum_bf16_kernel, dim3(4 * L2->units), dim3(256), 0, st, (const bf16_t*)L2->dzT_t, L2->ld_t, 0, T * B, L2->db_p);
    if (L1->db_p) hipLaunchKernelGGL(pst_rowsum_bf16_kernel, dim3(4 * L1->units), dim3(256), 0, st, (const bf16_t*)L1->dzT_t, L1->ld_t, 0, T * B, L1->db_p);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
