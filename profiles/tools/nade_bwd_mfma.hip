// NADE backward on the matrix cores (gfx950): the reverse scan of nade.py:199-229's autodiff with the sums over ROWS
// taken inside MFMAs (rows are the K dimension) instead of a cross-wave exchange per 4 visibles.
//
//   per row, visibles i = D-1 .. 0, state a_{i+1} = a_i + v_i w_enc[i], h = sigmoid(a):
//     d w_dec[i] += dl_i h_i ;  c += dl_i w_dec[i]  (per constant-h segment) ;  at v_i = 1: G += c h(1-h), d w_enc[i] += G, c = 0, a -= w_enc[i]
//   d b_enc = G_0.
//
// Layout.  A WAVE owns 32 rows x 32 hidden units; its state tiles a, h, c, G live in the C/D layout of v_mfma_f32_32x32x16 (lane = hidden unit
// n = lane & 31, register e <-> row (e & 3) + 8 (e >> 2) + 4 (lane >> 5)): 64 registers.  A workgroup = 8 waves = 256 rows of ONE 32-unit slice;
// the visibles are walked in blocks of 32, top block first.  Per block b (visibles i0 .. i0 + 31) and wave:
//   c   += DL . Wd_b                  A = dl [32 rows x 32 vis] (f16, loaded row-major: the A layout), B = w_dec fragments (pre-packed f16)
//   X    = DL . I                      the same dl as a C-layout tile [row][vis] (identity product: exact) -> the A operand of X^T . B
//   dWd += X^T . H                     rows as K: the state tile itself is the B operand (k-permuted the same way as X; cdna_hip_programming.md
//                                      "An accumulator tile as the next MFMA's operand")
// and for the rows with v = 1 inside the block, one RANK at a time (rank k = the k-th flip of a row counted from the top of the block; the
// loop runs max-flips-per-row times, ~4 at rho = 0.03):
//   W    = DL_{i <= f} . Wd_b          prefix-masked dl (a 128-bit and-mask per fragment from a 9-entry LDS table)
//   rows with a rank-k flip at f:  seg = c - W ; G += seg h(1-h) ; c = W ; a -= w_enc[f] ; h' = sigmoid(a)
//   dWe += OneHot^T . G                OneHot[row][vis] = (vis == f): built directly in the C layout (lane = visible), G = the rows' new G
//   dWd += X_{i <= f}^T . (h' - h)     the block's d w_dec uses the state of each visible: entry state + corrections below every flip
// The per-register work is skipped with one scalar branch where neither lane half's row has a flip of that rank.  At the end of a block the
// 8 waves' [32 vis x 32 hid] partial tiles of d w_dec / d w_enc meet in LDS (64 KB) and leave as ONE f32 atomic per (visible, hidden unit)
// and 256 rows: 4 x fewer atomic bytes than the 64-row workgroups of nade_bwd_kernel, whose 3.7 GB of adds are its floor at the chip's
// ~1.3 TB/s atomic rate (MI355X_MICROARCH.md "Global float atomics").
// Operands are IEEE half (dl carries the loss scale of the fp16 mode: |dl| <= 256), sums f32; the dense-input form (every other visible a
// flip) stays on nade_bwd_kernel (the density gate decides).
// PROTOTYPE, not part of the library (it loses to nade_bwd_kernel: profiles/round5_c_nade_bwd_mfma_notes.md).  To run it again: check out the
// commit "NADE backward on the matrix cores (rows as K): prototype wired ..." (header entries, loader signatures, ops wrappers, build list) and
// run profiles/tools/nade_bwd_probe.py there.  Ablation macros: NBM_NO_RANKS, NBM_NO_PAIRS, NBM_NO_EXCHANGE.
#include "../../multinn_amd/csrc/common.h"
#include <stdlib.h>

typedef float nb_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 nb_h8 __attribute__((ext_vector_type(8)));
typedef unsigned int nb_u4 __attribute__((ext_vector_type(4)));

#define NB_ROWS_WG 256
#define NB_EX_FLOATS (8 * 2 * 32 * 32)            // exchange: [wave][d w_dec | d w_enc][visible][hidden]
#define NB_WE_FLOATS (2 * 32 * 32)                // w_enc rows of the block, two buffers
#define NB_LDS_BYTES ((NB_EX_FLOATS + NB_WE_FLOATS) * 4 + 16 * 16 + 2 * 64 * 16)      // + the 9 and-masks + the identity fragments

__device__ __forceinline__ nb_f32x16 nb_mfma(nb_h8 a, nb_h8 b, nb_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ uint32_t nb_pack(float lo, float hi) { return (uint32_t)f32_to_f16(lo) | ((uint32_t)f32_to_f16(hi) << 16); }
__device__ __forceinline__ nb_h8 nb_frag(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    nb_u4 u = {a, b, c, d};
    return __builtin_bit_cast(nb_h8, u);
}

// w_dec f32 [tracks][D][HnT] -> B fragments of the products DL . Wd_b: wdp[((track * nslice + slice) * NB + block) * 2 + s][lane][8] (f16),
// element j of lane l = w_dec[32 block + 16 s + 8 (l >> 5) + j][32 slice + (l & 31)] (zero past D)
__global__ void __launch_bounds__(256) nade_bwd_pack_kernel(const float* __restrict__ wd, int tracks, int D, int HnT, int NB, uint16_t* __restrict__ wdp) {
    const int nslice = HnT / 32;
    const long total = (long)tracks * nslice * NB * 2 * 64 * 8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = idx & 7, l = (idx >> 3) & 63, s = (idx >> 9) & 1;
        long q = idx >> 10;
        const int b = q % NB; q /= NB;
        const int sl = q % nslice, m = q / nslice;
        const int i = 32 * b + 16 * s + 8 * (l >> 5) + j, hid = 32 * sl + (l & 31);
        wdp[idx] = i < D ? f32_to_f16(wd[((size_t)m * D + i) * HnT + hid]) : (uint16_t)0;
    }
}

__global__ void __launch_bounds__(512, 2)
nade_bwd_mfma_kernel(int tracks, int N, int D, int HnT, int nrg, const uint8_t* __restrict__ v, long v_track_stride, int ld_bias,
                     const float* __restrict__ w_enc, const uint16_t* __restrict__ wdp, const float* __restrict__ a_final,
                     float* __restrict__ d_bias, float* __restrict__ d_w_enc, float* __restrict__ d_w_dec, const int* __restrict__ run_if, int run_val) {
    extern __shared__ __attribute__((aligned(16))) char nb_smem[];
    if (run_if != nullptr && *run_if != run_val) return;                      // density gate (uniform)
    float* ex = reinterpret_cast<float*>(nb_smem);
    float* we_lds = ex + NB_EX_FLOATS;
    nb_u4* mtab = reinterpret_cast<nb_u4*>(we_lds + NB_WE_FLOATS);            // mtab[t] = the low 16 t bits of 128 set, t = 0 .. 8
    const int nslice = HnT >> 5;
    // the slices of one row group run back to back on one XCD (blocks b and b + 8 share one): dl and v are fetched into that L2 once
    const int bid = blockIdx.x, q = bid >> 3;
    const int sl = q % nslice, rg = (q / nslice) * 8 + (bid & 7);
    if (rg >= nrg) return;
    const int m = blockIdx.y, hb = sl * 32;
    const int NB = (D + 31) >> 5;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int R0 = rg * NB_ROWS_WG + w * 32;
    const int rowA = R0 + n;                                                  // the row this lane loads dl / v for (A layout)
    const bool rowA_ok = rowA < N;
    const int rowAc = rowA_ok ? rowA : N - 1;
    const uint8_t* __restrict__ vrow = v + (size_t)m * v_track_stride + (size_t)rowAc * D;
    const float* __restrict__ dlrow = d_bias + (size_t)rowAc * ld_bias + tracks * HnT + m * D;
    const float* __restrict__ we = w_enc + (size_t)m * D * HnT + hb;
    const nb_h8* __restrict__ wfrag = reinterpret_cast<const nb_h8*>(wdp) + ((size_t)(m * nslice + sl) * NB) * 2 * 64 + lane;
    if (threadIdx.x < 9) {
        const int t = threadIdx.x;
        nb_u4 mk4;
#pragma unroll
        for (int d = 0; d < 4; ++d) { const int kb = min(max(16 * t - 32 * d, 0), 32); mk4[d] = kb >= 32 ? 0xffffffffu : ((1u << kb) - 1u); }
        mtab[t] = mk4;
    }
    // identity fragments (B operand, k = 16 s + 8 hh + j, column n) live in LDS: 8 registers less across the block loop
    nb_h8* idtab = reinterpret_cast<nb_h8*>(mtab + 16);
    if (w < 2) {
        nb_h8 idf;
#pragma unroll
        for (int j = 0; j < 8; ++j) idf[j] = (16 * w + 8 * hh + j == n) ? (_Float16)1.0f : (_Float16)0.0f;
        idtab[w * 64 + lane] = idf;
    }

    // state tiles (C layout)
    float a[16], h[16], c[16], G[16];       // scalars, not vector types: a conditional element update of a vector value copies the whole vector
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = min(R0 + (e & 3) + 8 * (e >> 2) + 4 * hh, N - 1);
        a[e] = a_final[((size_t)m * N + row) * HnT + hb + n];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) { h[e] = fast_sigmoid(a[e]); c[e] = 0.f; G[e] = 0.f; }

    // loads of one block for this lane: dl (A layout: 16 visibles of its row), the row's v bytes (16 of the 32), the w_dec fragments.
    // TAIL: the top block may be partial (D % 4 == 0: a piece of 4 is inside or outside as a whole; outside pieces read index 0 and are masked)
    float4 dlr[2][2];
    uint32_t vb[4];
    nb_h8 wdf[2];
    auto load_blk = [&](int b, bool tail) {
        const float* __restrict__ pd = dlrow + 32 * b + 8 * hh;
        const uint8_t* __restrict__ pv = vrow + 32 * b + 16 * hh;
        if (!tail) {                                                           // one base address per stream, immediate offsets
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) dlr[s][p] = *reinterpret_cast<const float4*>(pd + 16 * s + 4 * p);
#pragma unroll
            for (int k = 0; k < 4; ++k) vb[k] = *reinterpret_cast<const uint32_t*>(pv + 4 * k);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int i = 32 * b + 16 * s + 8 * hh + 4 * p;
                    dlr[s][p] = *reinterpret_cast<const float4*>(dlrow + (i < D ? i : 0));
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 32 * b + 16 * hh + 4 * k;
                vb[k] = *reinterpret_cast<const uint32_t*>(vrow + (i < D ? i : 0));
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) wdf[s] = wfrag[((size_t)b * 2 + s) * 64];
    };
    // this thread's two w_enc values of a block -> LDS [visible][hidden]
    const int wv_vis = threadIdx.x >> 4, wv_hid = (threadIdx.x & 15) * 2;
    auto load_we = [&](int b) -> float2 {
        const int i = 32 * b + wv_vis;
        return *reinterpret_cast<const float2*>(we + (size_t)(i < D ? i : 0) * HnT + wv_hid);
    };
    load_blk(NB - 1, true);
    {
        const float2 w0 = load_we(NB - 1);
        *reinterpret_cast<float2*>(we_lds + ((NB - 1) & 1) * 1024 + wv_vis * 32 + wv_hid) = w0;
    }
    __syncthreads();

    for (int b = NB - 1; b >= 0; --b) {
        const int i0 = 32 * b;
        const float* __restrict__ wel = we_lds + (b & 1) * 1024;

        // ---- this lane's operands of the block (A layout: row = lane & 31, visibles 16 s + 8 hh + j)
        nb_h8 A[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t pk[4];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bool ok = rowA_ok && (i0 + 16 * s + 8 * hh + 4 * p < D);
                const float4 d = dlr[s][p];
                pk[2 * p] = ok ? nb_pack(d.x, d.y) : 0u;
                pk[2 * p + 1] = ok ? nb_pack(d.z, d.w) : 0u;
            }
            A[s] = nb_frag(pk[0], pk[1], pk[2], pk[3]);
        }
        uint32_t m16 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = rowA_ok && (i0 + 16 * hh + 4 * k < D);
            const uint32_t nib = ((vb[k] * 0x01020408u) >> 24) & 0xfu;    // bytes are 0 / 1: byte t of the word -> bit t
            m16 |= (ok ? nib : 0u) << (4 * k);
        }
        uint32_t maskA;                                                       // the row's 32 bits (both lane halves hold them)
        {
            const auto r = __builtin_amdgcn_permlane32_swap(m16, m16, false, false);
            maskA = r[0] | (r[1] << 16);
        }
        // ---- dense part
        nb_f32x16 dwd, dwe;
        uint32_t Xp[8];                                                       // X = dl as a C-layout tile [row][visible], packed pairs (registers 2 p, 2 p + 1)
        {
            nb_f32x16 X;
#pragma unroll
            for (int e = 0; e < 16; ++e) { X[e] = 0.f; dwd[e] = 0.f; dwe[e] = 0.f; }
            nb_f32x16 cv;
#pragma unroll
            for (int e = 0; e < 16; ++e) cv[e] = c[e];
            cv = nb_mfma(A[0], wdf[0], cv);
            cv = nb_mfma(A[1], wdf[1], cv);
#pragma unroll
            for (int e = 0; e < 16; ++e) c[e] = cv[e];
            X = nb_mfma(A[0], idtab[lane], X);
            X = nb_mfma(A[1], idtab[64 + lane], X);
#pragma unroll
            for (int p = 0; p < 8; ++p) Xp[p] = nb_pack(X[2 * p], X[2 * p + 1]);
        }
        {
            uint32_t Hp[8];                                                   // the entry state as the B operand (fragment s = pairs 4 s .. 4 s + 3)
#pragma unroll
            for (int p = 0; p < 8; ++p) Hp[p] = nb_pack(h[2 * p], h[2 * p + 1]);
            dwd = nb_mfma(nb_frag(Xp[0], Xp[1], Xp[2], Xp[3]), nb_frag(Hp[0], Hp[1], Hp[2], Hp[3]), dwd);
            dwd = nb_mfma(nb_frag(Xp[4], Xp[5], Xp[6], Xp[7]), nb_frag(Hp[4], Hp[5], Hp[6], Hp[7]), dwd);
        }

        // ---- the flips, one rank at a time (rank = the k-th flip of a row counted from the top of the block)
#ifdef NBM_NO_RANKS
        maskA = 0u;
#endif
        while (__builtin_amdgcn_ballot_w64(maskA != 0u) != 0ull) {
            const int fA = maskA != 0u ? 31 - __builtin_clz(maskA) : -1;     // this row's flip of the rank (position in the block), -1: none
            if (maskA != 0u) maskA &= ~(1u << fA);
            // the C layout's rows: register e of this lane half <-> row (e & 3) + 8 (e >> 2) + 4 hh, whose f the lane of that row holds
            int fe[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) fe[e] = __builtin_amdgcn_ds_bpermute(4 * ((e & 3) + 8 * (e >> 2) + 4 * hh), fA);
            nb_f32x16 W;
#pragma unroll
            for (int e = 0; e < 16; ++e) W[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int t = min(max(fA + 1 - 16 * s - 8 * hh, 0), 8);       // how many of this fragment's 8 visibles are <= f
                const nb_u4 am = mtab[t];
                const nb_u4 au = __builtin_bit_cast(nb_u4, A[s]);
                W = nb_mfma(nb_frag(au[0] & am[0], au[1] & am[1], au[2] & am[2], au[3] & am[3]), wdf[s], W);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t Gp[4], Dp[4], Op[4], Vp[4];                          // B: new G of the flipping rows, h' - h; A: one-hot, prefix-masked X
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int p = 4 * s + pp;
                    Gp[pp] = 0u; Dp[pp] = 0u; Op[pp] = 0u; Vp[pp] = 0u;
#ifdef NBM_NO_PAIRS
                    continue;
#endif
                    if (__builtin_amdgcn_ballot_w64((fe[2 * p] & fe[2 * p + 1]) >= 0) == 0ull) continue;      // no row of this register pair flips (scalar)
                    float gv[2], dv[2];
                    uint32_t om = 0u, vm = 0u;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int e = 2 * p + t;
                        const bool pe = fe[e] >= 0;
                        const int f = max(fe[e], 0);
                        const float seg = c[e] - W[e];
                        const float Gn = fmaf(seg, fmaf(-h[e], h[e], h[e]), G[e]);
                        const float an = a[e] - wel[f * 32 + n];
                        const float hn = fast_sigmoid(an);
                        gv[t] = pe ? Gn : 0.f;
                        dv[t] = pe ? hn - h[e] : 0.f;
                        G[e] = pe ? Gn : G[e];
                        c[e] = pe ? W[e] : c[e];
                        a[e] = pe ? an : a[e];
                        h[e] = pe ? hn : h[e];
                        const uint32_t half = t ? 0xffff0000u : 0x0000ffffu;
                        om |= (fe[e] == n) ? (0x3c003c00u & half) : 0u;       // 1.0h at (row, visible f)
                        vm |= (n <= fe[e]) ? half : 0u;
                    }
                    Gp[pp] = nb_pack(gv[0], gv[1]);
                    Dp[pp] = nb_pack(dv[0], dv[1]);
                    Op[pp] = om;
                    Vp[pp] = Xp[p] & vm;
                }
                dwe = nb_mfma(nb_frag(Op[0], Op[1], Op[2], Op[3]), nb_frag(Gp[0], Gp[1], Gp[2], Gp[3]), dwe);
                dwd = nb_mfma(nb_frag(Vp[0], Vp[1], Vp[2], Vp[3]), nb_frag(Dp[0], Dp[1], Dp[2], Dp[3]), dwd);
            }
        }

        // ---- the next block's operands are requested HERE (not at the top of the block: 30 registers less across the rank loop); the
        // exchange below covers their flight
        float2 wnext = make_float2(0.f, 0.f);
        if (b > 0) { load_blk(b - 1, false); wnext = load_we(b - 1); }
        // ---- the 8 waves' partial tiles meet in LDS; one f32 atomic per (visible, hidden unit) and workgroup
        float* exw = ex + w * 2048;
#ifndef NBM_NO_EXCHANGE
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int vis = (e & 3) + 8 * (e >> 2) + 4 * hh;
            exw[vis * 32 + n] = dwd[e];
            exw[1024 + vis * 32 + n] = dwe[e];
        }
        __syncthreads();
        {
            const int mat = threadIdx.x >> 8, rem = threadIdx.x & 255;
            const int visq = rem >> 5, hid = rem & 31;
            float* __restrict__ dst = (mat ? d_w_enc : d_w_dec) + ((size_t)m * D + i0) * HnT + hb + hid;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int vis = visq * 4 + j;
                float sum = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) sum += ex[ww * 2048 + mat * 1024 + vis * 32 + hid];
                if (i0 + vis < D && (mat == 0 || sum != 0.f)) atomicAdd(dst + (size_t)vis * HnT, sum);
            }
        }
#else
        asm volatile("" :: "v"(dwd), "v"(dwe), "v"(exw));
#endif
        if (b > 0) *reinterpret_cast<float2*>(we_lds + ((b - 1) & 1) * 1024 + wv_vis * 32 + wv_hid) = wnext;
        __syncthreads();
    }
    // close the last open segment; d b_enc = G_0
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = R0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        const float g = fmaf(c[e], fmaf(-h[e], h[e], h[e]), G[e]);
        if (row < N) d_bias[(size_t)row * ld_bias + m * HnT + hb + n] = g;
    }
}

extern "C" int mnn_nade_bwd_mfma_ok(int D, int Hn) { return (D > 0 && D % 4 == 0 && Hn >= 32 && Hn <= 256 && Hn % 32 == 0) ? 1 : 0; }

extern "C" size_t mnn_nade_bwd_pack_bytes(int tracks, int D, int Hn) { return (size_t)tracks * ((D + 31) / 32) * 32 * Hn * 2; }

extern "C" int mnn_nade_bwd_pack(mnn_stream_t s, int tracks, int D, int Hn, const float* w_dec, void* wdp) {
    MNN_REQUIRE(mnn_nade_bwd_mfma_ok(D, Hn), "mnn_nade_bwd_pack: needs D %% 4 == 0 and Hn a multiple of 32 up to 256 (D=%d Hn=%d)", D, Hn);
    MNN_REQUIRE(tracks > 0 && w_dec && wdp && ((uintptr_t)wdp & 15) == 0, "mnn_nade_bwd_pack: null / unaligned pointer");
    const int NB = (D + 31) / 32;
    const long total = (long)tracks * (Hn / 32) * NB * 2 * 64 * 8;
    hipLaunchKernelGGL(nade_bwd_pack_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)s, w_dec, tracks, D, Hn, NB,
                       (uint16_t*)wdp);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_nade_logprob_bwd_mfma(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride, int ld_bias,
                                         const float* w_enc, const void* wdp, const float* a_final, float* d_bias, float* d_w_enc, float* d_w_dec,
                                         const int* run_if, int run_val) {
    MNN_REQUIRE(mnn_nade_bwd_mfma_ok(D, Hn), "mnn_nade_logprob_bwd_mfma: needs D %% 4 == 0 and Hn a multiple of 32 up to 256 (D=%d Hn=%d)", D, Hn);
    MNN_REQUIRE(tracks > 0 && N > 0 && v && w_enc && wdp && a_final && d_bias && d_w_enc && d_w_dec, "mnn_nade_logprob_bwd_mfma: null pointer / empty problem");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D) && ld_bias % 4 == 0 && ((uintptr_t)d_bias & 15) == 0 && ((uintptr_t)v & 3) == 0 && v_track_stride % 4 == 0 &&
                    ((uintptr_t)w_enc & 7) == 0 && ((uintptr_t)wdp & 15) == 0,
                "mnn_nade_logprob_bwd_mfma: ld_bias (%d) must be a multiple of 4 covering tracks * (Hn + D); d_bias 16-byte, v 4-byte, w_enc 8-byte aligned", ld_bias);
    hipStream_t st = (hipStream_t)s;
    static bool attr_set[64];
    int dev = 0;
    MNN_HIP(hipGetDevice(&dev));
    MNN_REQUIRE(dev >= 0 && dev < 64, "mnn_nade_logprob_bwd_mfma: device index %d", dev);
    if (!attr_set[dev]) {
        MNN_HIP(hipFuncSetAttribute((const void*)nade_bwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NB_LDS_BYTES));
        attr_set[dev] = true;
    }
    const int nrg = cdiv(N, NB_ROWS_WG), nslice = Hn / 32;
    hipLaunchKernelGGL(nade_bwd_mfma_kernel, dim3(cdiv(nrg, 8) * 8 * nslice, tracks), dim3(512), NB_LDS_BYTES, st, tracks, N, D, Hn, nrg, v, v_track_stride, ld_bias,
                       w_enc, (const uint16_t*)wdp, a_final, d_bias, d_w_enc, d_w_dec, run_if, run_val);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
