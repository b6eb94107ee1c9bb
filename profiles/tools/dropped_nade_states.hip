// NADE log-prob forward, STATE-MAJOR form on the matrix cores (Hn = 256, D <= 512; bf16 operands, f32 accumulation).
// Reference: /root/reference/multinn/models/common/nade.py:155-229.
//
// a_{d+1} = a_d + v_d w_enc[d] only moves at visibles with v = 1, so a row has 1 + nnz(v) hidden STATES; state k of a row
// serves the visibles (j_k, j_{k+1}] between two of its flips.  nade_mfma.hip walks column tiles and, inside a tile, passes
// over the flips (three barriers and a few hundred VALU instructions around 16 small MFMAs per tile: issue- and
// latency-bound).  Here the loop nest is turned inside out:
//   * a workgroup (8 waves) is persistent over a contiguous range of rows and keeps ALL of w_dec in registers as MFMA B
//     fragments: wave w owns the column tiles 2w, 2w+1 (2 x 16 k-steps x 4 VGPRs = 128 VGPRs), loaded once;
//   * rows are taken in groups of up to 16 (as many as give <= 512 states); their states are listed in LDS in (row, k) order:
//     source of the increment (the row's bias c, or w_enc[j] of the flip that creates the state) and the visibles [lo, hi) served;
//   * per chunk of 32 consecutive states:  phase 1 (thread = hidden unit): a = running sum with reset at a row's first
//     state (f32, exact order of nade.py), h = sigmoid(a) -> bf16 state tile in LDS;  phase 2: [32 states x 256] x w_dec^T
//     for ALL column tiles, 32 v_mfma_f32_32x32x16_bf16 per wave back to back, no K split, no cross-wave reduction;
//     phase 3: every accumulator element whose column lies in its state's [lo, hi) IS the logit of that (row, visible): add
//     b_dec, finish p, the NLL term and d nll / d b_dec; (state, tile) pairs that do not intersect are skipped wave-uniformly.
// ~7x more MFMA work than the tile form (every state against every column) -- about 1/4 of the kernel's time -- for two
// barriers per 32 states and no per-flip bookkeeping.  a_final (the last state's pre-activation) goes to the backward pass.
#include "common.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#ifdef NS_TRACE      // development only: per-phase clocks of one wave (scratch/nade_states_trace.hip)
__device__ long long ns_trace[16];
#define NS_T(k) do { if (tid == NS_TRACE_TID && blockIdx.x == NS_TRACE) { const long long now_ = wall_clock64(); ns_trace[k] += now_ - tprev_; tprev_ = now_; } } while (0)
#else
#define NS_T(k) do { } while (0)
#endif
#define NADE_EPS 1e-6f
#define LN2F 0.6931471805599453f
#define NS_H 256
#define NS_PITCH 264          // bf16 elements per state-tile row (rows 16-B aligned, 4-bank skew)
#define NS_ROWS 16            // rows per group, at most
#define NS_DMAX 512
#define NS_SMAX 512           // states per group (a group of one row may exceed it: D + 1 <= 513)
#define NS_SCAP 544           // list capacity: 513 rounded up to a multiple of 32

__device__ __forceinline__ void ns_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float ns_ln(float x) { return __builtin_amdgcn_logf(x) * LN2F; }

struct NsSmem {
    bf16_t sH[2][32][NS_PITCH];         // the states of chunk c (buffer c & 1): written while chunk c-1 is multiplied
    float sLg[NS_ROWS][NS_DMAX];        // logits of the group's (row, visible) pairs, then their log-prob terms
    unsigned sV[NS_ROWS][NS_DMAX / 32]; // v bits
    int sBase[NS_ROWS + 1];             // first state of each row; [nrows] = number of states
    unsigned short sSrc[NS_SCAP];       // 0xFFFF: the row's first state (a = c); else the flip j that creates the state (a += w_enc[j])
    unsigned sSeg[NS_SCAP];             // visibles served [lo, hi), row, last-state flag: lo | hi << 10 | row << 20 | last << 24
    int nrows, nstates;
};

// Column tiles per wave: waves 0-3 also produce the states (VALU: running sums + sigmoids), so they multiply one tile each
// (tiles 0..3) and waves 4-7 three each (tiles 4..15): every SIMD hosts one wave of each kind, MFMA of one overlaps VALU of the other.
__global__ void __launch_bounds__(512)
nade_fwd_states_kernel(int tracks, int N, int D, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias, int ld_bias,
                       const float* __restrict__ w_enc, const bf16_t* __restrict__ w_dec_bf, const float* __restrict__ row_weight,
                       float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias, float* __restrict__ a_final) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    NsSmem& S = *reinterpret_cast<NsSmem*>(smem_raw);
    constexpr int Hn = NS_H;
    const int m = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 31, hh = lane >> 5;
    const bool producer = w < 4;
    const int j = tid & 255;                                  // producer waves: the hidden unit of this thread
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const bf16_t* __restrict__ wd = w_dec_bf + (size_t)m * D * Hn;
    const float* __restrict__ cb = bias + m * Hn;             // + row * ld_bias: the row's hidden bias c
    const int bd_off = tracks * Hn + m * D;
    const int ntile = (D + 31) / 32;
    const int nw64 = (D + 63) / 64;
    const int t_first = producer ? w : 4 + 3 * (w - 4);
    const int ntl = max(0, min(producer ? 1 : 3, ntile - t_first));      // tiles of this wave

    // w_dec as MFMA B fragments, resident for the whole launch: lane -> column 32 t + (l & 31), k = 16 ks + 8 (l >> 5) + 0..7
    bf16x8_t Bf[3][16];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q < ntl) {
            const int d = min(32 * (t_first + q) + lc, D - 1);
            const bf16_t* p = wd + (size_t)d * Hn + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) Bf[q][ks] = *reinterpret_cast<const bf16x8_t*>(p + 16 * ks);
        }
    }

    for (int s = tid; s < NS_SCAP; s += 512) { S.sSeg[s] = 0u; S.sSrc[s] = 0xFFFFu; }     // the lists are OR-ed together: start (and stay) clear
    const int r_begin = (int)((long)blockIdx.x * N / gridDim.x), r_end = (int)((long)(blockIdx.x + 1) * N / gridDim.x);
#ifdef NS_TRACE
    long long tprev_ = wall_clock64();
#endif
    for (int r0 = r_begin; r0 < r_end;) {
        const int nc = min(NS_ROWS, r_end - r0);
        NS_T(0);
        // ---- v bits of the candidate rows: wave w takes rows 2w, 2w+1 ----
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * w + rr;
            unsigned char vb[NS_DMAX / 64];
#pragma unroll
            for (int cw = 0; cw < NS_DMAX / 64; ++cw) {
                const int col = 64 * cw + lane;
                vb[cw] = (row < nc && cw < nw64 && col < D) ? vm[(size_t)(r0 + row) * D + col] : (unsigned char)0;
            }
#pragma unroll
            for (int cw = 0; cw < NS_DMAX / 64; ++cw) {
                const unsigned long long bal = __ballot(vb[cw] != 0);
                if (lane == 0) { S.sV[row][2 * cw] = (unsigned)bal; S.sV[row][2 * cw + 1] = (unsigned)(bal >> 32); }
            }
        }
        NS_T(1);
        ns_barrier();
        NS_T(2);
        // ---- how many of them fit (<= NS_SMAX states, at least one row); first state of each ----
        if (w == 0) {
            int cnt = 0;
            if (lane < nc) {
                cnt = 1;
#pragma unroll
                for (int k = 0; k < NS_DMAX / 32; ++k) cnt += __popc(S.sV[lane][k]);
            }
            int inc = cnt;                                   // inclusive prefix over lanes 0..15
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                const int t = __shfl_up(inc, o);
                if (lane >= o) inc += t;
            }
            const bool fits = lane < nc && (lane == 0 || inc <= NS_SMAX);
            const unsigned long long fb = __ballot(fits);
            const int nrows = __builtin_ctzll(~fb);          // rows 0 .. nrows-1 fit (prefix property: inc is monotone)
            if (lane < NS_ROWS) S.sBase[lane] = inc - cnt;
            if (lane == nrows - 1) { S.sBase[nrows] = inc; S.nstates = inc; S.nrows = nrows; }
        }
        ns_barrier();
        NS_T(3);
        const int nrows = S.nrows, nstates = S.nstates;
        const int nch = (nstates + 31) / 32;
        // ---- state lists: wave w lists rows 2w, 2w+1; lane = one 32-bit word of the row's v bits, slots by popcount prefix ----
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * w + rr;
            if (row < nrows) {
                const unsigned bits0 = lane < NS_DMAX / 32 ? S.sV[row][lane] : 0u;
                int inc = __popc(bits0);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const int t = __shfl_up(inc, o);
                    if (lane >= o) inc += t;
                }
                // state k >= 1 of the row is created by the row's k-th flip; state 0 by the bias.  Entry k: src = flip k (or bias),
                // lo = flip k + 1 (or 0), hi = flip (k+1) + 1 (or D), last = no later flip.
                const int base = S.sBase[row];
                int k = inc - __popc(bits0);                 // flips before this word
                unsigned bits = bits0;
                // the previous flip (for lo/src of the state that a flip of this word CLOSES) travels with the loop:
                // entry k's hi is set by flip k+1, entry k+1's lo/src by the same flip.
                while (bits) {
                    const unsigned jf = 32u * lane + __builtin_ctz(bits);
                    bits &= bits - 1;
                    // flip number k+1 (1-based) at visible jf: closes state k, opens state k+1
                    atomicOr(&S.sSeg[base + k], (jf + 1) << 10);                        // hi of state k
                    S.sSrc[base + k + 1] = (unsigned short)jf;
                    atomicOr(&S.sSeg[base + k + 1], (jf + 1) | ((unsigned)row << 20));  // lo, row of state k+1
                    ++k;
                }
                const int nfl = __shfl(inc, 15);             // flips of the row
                if (lane == 0) {
                    S.sSrc[base] = 0xFFFFu;
                    atomicOr(&S.sSeg[base], (unsigned)row << 20);
                    atomicOr(&S.sSeg[base + nfl], ((unsigned)D << 10) | (1u << 24));    // the last state serves up to D
                }
            }
        }
        NS_T(4);
        ns_barrier();
        NS_T(5);

        // ---- chunks of 32 states ----
        float acur = 0.f;
        float x[32];
        unsigned my_src = 0, my_seg = 0;
        auto fetch = [&](int ch) {                            // producer waves: descriptors and increments of chunk ch
            my_src = S.sSrc[32 * ch + lc]; my_seg = S.sSeg[32 * ch + lc];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int src = __builtin_amdgcn_readlane((int)my_src, u);
                const int row = (__builtin_amdgcn_readlane((int)my_seg, u) >> 20) & 15;
                const float* p = src == 0xFFFF ? cb + (size_t)(r0 + row) * ld_bias : we + (size_t)src * Hn;
                x[u] = p[j];
            }
        };
        auto produce = [&](int ch) {                          // running sums (reset at a row's first state) -> sigmoid -> bf16 state tile
            bf16_t (*H)[NS_PITCH] = S.sH[ch & 1];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int src = __builtin_amdgcn_readlane((int)my_src, u);
                acur = src == 0xFFFF ? x[u] : acur + x[u];
                H[u][j] = f32_to_bf16(fast_sigmoid(acur));
                const int sg = __builtin_amdgcn_readlane((int)my_seg, u);
                if (a_final != nullptr && (sg >> 24) != 0) a_final[((size_t)m * N + r0 + ((sg >> 20) & 15)) * Hn + j] = acur;
            }
        };
        if (producer) { fetch(0); produce(0); }
        ns_barrier();
        for (int ch = 0; ch < nch; ++ch) {
            const int base = 32 * ch;
            NS_T(6);
            if (producer && ch + 1 < nch) fetch(ch + 1);     // the next chunk's increments fly under this chunk's MFMAs
            unsigned seg[16];                                // descriptors of the 16 states this lane's accumulator rows belong to
#pragma unroll
            for (int e = 0; e < 16; ++e) seg[e] = S.sSeg[base + (e & 3) + 8 * (e >> 2) + 4 * hh];
            if (ntl > 0) {
                f32x16_t acc[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
                const bf16_t* ap = &S.sH[ch & 1][lc][8 * hh];
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(ap + 16 * ks);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[0][ks], acc[0], 0, 0, 0);
                    if (ntl > 1) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[1][ks], acc[1], 0, 0, 0);
                    if (ntl > 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, Bf[2][ks], acc[2], 0, 0, 0);
                }
                NS_T(7);
                // the accumulator elements whose column lies in their state's [lo, hi) ARE the logits of (row, column)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (q < ntl) {
                        const int col = 32 * (t_first + q) + lc;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int lo = seg[e] & 1023, hi = (seg[e] >> 10) & 1023;
                            if (col >= lo && col < hi) S.sLg[(seg[e] >> 20) & 15][col] = acc[q][e];
                        }
                    }
                }
            }
            NS_T(8);
            if (producer && ch + 1 < nch) produce(ch + 1);
            NS_T(9);
            ns_barrier();                                    // chunk ch consumed, chunk ch+1 produced
            NS_T(10);
        }
        // ---- pointwise of the group's (row, visible) pairs, dense: p, the NLL term, d nll / d b_dec ----
        for (int idx = tid; idx < nrows * D; idx += 512) {
            const int row = idx / D, col = idx - row * D, grow = r0 + row;
            const float l = S.sLg[row][col] + bias[(size_t)grow * ld_bias + bd_off + col];
            const bool on = (S.sV[row][col >> 5] >> (col & 31)) & 1u;
            const float pr = fast_sigmoid(l);
            const float qr = fast_sigmoid(-l);               // 1-p without cancellation
            S.sLg[row][col] = on ? ns_ln(NADE_EPS + pr) : ns_ln(NADE_EPS + qr);
            if (cond_p != nullptr) cond_p[((size_t)m * N + grow) * D + col] = pr;
            if (d_bias != nullptr) {
                const float dnll_dp = on ? -fast_rcp(NADE_EPS + pr) : fast_rcp(NADE_EPS + qr);
                d_bias[(size_t)grow * ld_bias + bd_off + col] = row_weight[grow] * dnll_dp * pr * qr;
            }
        }
        ns_barrier();
        // ---- per-row NLL of the group: wave w sums rows 2w, 2w+1; the lists are cleared for the next group ----
        if (nll != nullptr) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = 2 * w + rr;
                if (row < nrows) {
                    float sum = 0.f;
                    for (int col = lane; col < D; col += 64) sum += S.sLg[row][col];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                    if (lane == 0) nll[(size_t)m * N + r0 + row] = -sum;
                }
            }
        }
        for (int s = tid; s < 32 * nch; s += 512) { S.sSeg[s] = 0u; S.sSrc[s] = 0xFFFFu; }
        NS_T(11);
        ns_barrier();                                        // before the next group overwrites the lists
        r0 += nrows;
    }
}

extern "C" int mnn_nade_states_ok(int D, int Hn) { return (Hn == NS_H && D > 0 && D <= NS_DMAX) ? 1 : 0; }

// Launch helper for nade_mfma.hip's entry point (same contract as mnn_nade_logprob_fwd_mfma).
int nade_fwd_states_launch(hipStream_t st, int tracks, int N, int D, const uint8_t* v, long v_track_stride, const float* bias, int ld_bias,
                           const float* w_enc, const bf16_t* w_dec_bf, const float* row_weight, float* nll, float* cond_p, float* d_bias,
                           float* a_final) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)nade_fwd_states_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(NsSmem));
    MNN_HIP(attr);
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        MNN_HIP(hipGetDevice(&dev));
        MNN_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // one persistent workgroup per CU (LDS and registers allow no second one), fewer when there are not enough 16-row groups
    int nbx = cus / tracks;
    if (nbx < 1) nbx = 1;
    nbx = min(nbx, cdiv(N, NS_ROWS));
    hipLaunchKernelGGL(nade_fwd_states_kernel, dim3(nbx, tracks), dim3(512), sizeof(NsSmem), st, tracks, N, D, v, v_track_stride, bias, ld_bias,
                       w_enc, w_dec_bf, row_weight, nll, cond_p, d_bias, a_final);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
