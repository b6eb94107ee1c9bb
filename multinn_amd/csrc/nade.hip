// NADE visible-order conditional scan for gfx950: log_prob forward, reverse-scan backward, sampling.
// Reference: /root/reference/multinn/models/common/nade.py:155-229 (log_prob), 231-308 (sample).
//
// Forward  (lane = hidden):  8 waves x 8 rows; h = sigmoid(a) cached and recomputed only at v = 1; the
//                            64 per-lane partial dot products of 8 rows x 8 visibles are reduced across
//                            lanes by one halving butterfly (permlane swaps + DPP), no LDS, no barrier.
// Backward (lane = hidden):  8 waves x 8 rows; the row is wave-uniform, so d l and v are scalars and the
//                            sums over rows for d w_dec / d w_enc accumulate in-lane; one LDS reduction
//                            + one 1 KiB-contiguous f32 atomic per visible per block.
// Sampling (one wave/row):   deterministic summation order + IEEE-only sigmoid so Bernoulli draws are
//                            bit-identical to oracle/det_ref.c.
#include "common.h"

#define NADE_EPS 1e-6f
#define LN2 0.6931471805599453f

__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * LN2; }

// Compacted ragged batches (mnn_ragged_index: the valid rows of a window first, its padding behind them): the scans take the number of valid
// rows from the DEVICE (n_rows_dev, may be NULL = all N rows) so that a captured step serves any lengths.  A workgroup whose first row lies
// beyond it has nothing to scan; it writes zeros where a later kernel reads this workgroup's rows (the Dense weight-gradient and input-gradient
// GEMMs run over all N rows) and leaves.  Rows of a partly valid workgroup are scanned like any other: their row weight is 0.
__device__ __forceinline__ bool nade_rows_beyond(const int* __restrict__ n_rows_dev, int first_row) {
    return n_rows_dev != nullptr && first_row >= *n_rows_dev;
}
__device__ __forceinline__ void nade_zero_rows(float* __restrict__ d_bias, int ld_bias, int col0, int ncol, int row0, int nrow, int N, int nthreads) {
    if (d_bias == nullptr) return;
    for (int e = threadIdx.x; e < nrow * ncol; e += nthreads) {
        const int n = e / ncol, i = e - n * ncol;
        if (row0 + n < N) d_bias[(size_t)(row0 + n) * ld_bias + col0 + i] = 0.f;
    }
}

// ----------------------------------------------------------------------------------------------
// forward (lane = hidden unit, 8 waves x 8 rows per block; no barrier, no cross-wave traffic)
//
// Sparsity (exact): `a` only changes at visibles with v = 1 (nade.py:219), so h = sigmoid(a) is cached in
// registers and recomputed only there (the row is wave-uniform, so that branch is scalar).  Per visible and
// row each lane forms its partial dot product over its HQ hidden units; the 8 rows x 8 visibles = 64
// partials per lane are then summed ACROSS the 64 lanes by a halving butterfly (v_permlane32_swap,
// v_permlane16_swap, DPP/swizzle xor steps): ~2.5 VALU ops per output instead of a 6-step reduction each,
// and lane L ends up owning the logit of (row L>>3, visible L&7).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_xor8(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false)); }
__device__ __forceinline__ float swz_xor4(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x101F)); }
// x[l ^ 4] without the LDS crossbar (a ds_swizzle is ~100 cycles on a dependent chain): two row shifts by four lanes, each written to the
// banks (groups of four lanes) it is right for -- row_shl:4 (lane l reads l + 4) into banks 0 and 2, row_shr:4 (l - 4) into banks 1 and 3
__device__ __forceinline__ float dpp_xor4(float x) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x104, 0xf, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, __float_as_int(x), 0x114, 0xf, 0xa, false);
    return __int_as_float(t);
}
__device__ __forceinline__ float dpp_xor2(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_xor1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false)); }

// Halving butterfly, applied in two parts.  rows8(): the 8 per-row partials of ONE visible -> one value per
// lane, lane bits 5..3 selecting the row (3 steps: permlane32_swap, permlane16_swap, xor-8).  vis8(): the 8
// such values of a chunk -> the lane's final sum, lane bits 2..0 selecting the visible (xor 4, 2, 1).
__device__ __forceinline__ float rows8(float (&x)[8], int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {        // lane bit 5 <-> row bit 2
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[k]), __float_as_uint(x[k + 4]), false, false);
        x[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {        // lane bit 4 <-> row bit 1
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x[k]), __float_as_uint(x[k + 2]), false, false);
        x[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const float t = x[0] + dpp_xor8(x[0]), u = x[1] + dpp_xor8(x[1]);   // lane bit 3 <-> row bit 0
    return (lane & 8) ? u : t;
}
__device__ __forceinline__ float vis8(float (&y)[8], int lane) {
    const bool b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = y[k] + swz_xor4(y[k]), u = y[k + 4] + swz_xor4(y[k + 4]);
        y[k] = b2 ? u : t;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float t = y[k] + dpp_xor2(y[k]), u = y[k + 2] + dpp_xor2(y[k + 2]);
        y[k] = b1 ? u : t;
    }
    const float t = y[0] + dpp_xor1(y[0]), u = y[1] + dpp_xor1(y[1]);
    return b0 ? u : t;
}

#define FWD_R 8
// Cooperative, double-buffered LDS staging of 8 consecutive rows of w_dec and w_enc (shared by the block's
// 8 waves): registers <- global for chunk c+1 while chunk c is consumed from LDS; one barrier per chunk.
template <int HQ>
struct WStage {
    static constexpr int W = HQ * 64;            // padded hidden width
    static constexpr int NE = (8 * W) / 512;     // elements per thread per matrix
    float rd[NE], re[NE];
    __device__ __forceinline__ void gload(const float* __restrict__ wd, const float* __restrict__ we, int i0, int D, int Hn, int ld = 0) {
        if (ld == 0) ld = Hn;                    // row stride (a hidden slice of a wider matrix passes its width as Hn)
#pragma unroll
        for (int k = 0; k < NE; ++k) {                      // unconditional loads from clamped (visible, hidden unit), zeroed afterwards: a
            const int e = threadIdx.x + k * 512;            // predicated load is a branch and a wait per element (see nade_bwd_kernel)
            const int ii = e / W, j = e - ii * W, i = i0 + ii;
            const bool ok = i >= 0 && i < D && j < Hn;
            const size_t o = (size_t)min(max(i, 0), D - 1) * ld + min(j, Hn - 1);
            const float xd = wd[o], xe = we[o];
            rd[k] = ok ? xd : 0.f;
            re[k] = ok ? xe : 0.f;
        }
    }
    __device__ __forceinline__ void lstore(float* __restrict__ sd, float* __restrict__ se) const {
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            sd[threadIdx.x + k * 512] = rd[k];
            se[threadIdx.x + k * 512] = re[k];
        }
    }
};

// UT (the density-gated launch of the 16-bit modes: DENSE batches): the hidden states are advanced MULTIPLICATIVELY.  With u = exp(-a),
// h = sigmoid(a) = 1 / (1 + u) and a flip a += w is u *= exp(-w): exp(-w[i]) is one exponential per (visible, hidden unit) for ALL the rows of the
// wave that flip there (at rho = 0.5 four of eight), and a flip itself is mul + add + rcp instead of add + mul + exp + add + rcp -- the scan of a
// dense batch is bound by exactly these transcendentals.  `a` is still summed exactly (it is handed to the backward pass), and it is what keeps u
// honest: a running max of |a| per lane is tested once per visible with flips, and above 40 every row's u is re-derived from its a before the
// next visible.  Below that bound a product of an u in [e^-40, e^40] and an exp(-w) in f32 range is exact to rounding whatever w is; past the
// f32 range it saturates to 0 / inf, where h is 1 / 0 to 1e-38 -- so the form is exact-safe for any weights.  The multiplies' rounding compounds
// to ~2e-6 relative over 220 flips (1e-4 is the mode's bound); the f32 parity entry point (mnn_nade_logprob_fwd) keeps the direct form.
template <int HQ, bool UT>
__global__ void __launch_bounds__(512)
nade_fwd_kernel(int tracks, int N, int D, int Hn, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias,
                int ld_bias, const float* __restrict__ w_enc, const float* __restrict__ w_dec, const float* __restrict__ row_weight,
                float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias, float* __restrict__ a_final,
                const int* __restrict__ gate, int run_if, const int* __restrict__ n_rows_dev, int* __restrict__ unsafe) {
    constexpr int W = HQ * 64;
    __shared__ float wl[2][2][8 * W];            // [buffer][w_dec | w_enc][visible-in-chunk][hidden]
    if (gate != nullptr && *gate != run_if) {           // density-gated pair of launches (mnn_nade_logprob_fwd_gated): uniform exit
        if (unsafe != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(unsafe, 1);      // not run = not vouched for
        return;
    }
    const int m = blockIdx.y;
    if (nade_rows_beyond(n_rows_dev, blockIdx.x * 64)) {                  // compacted ragged batch: all 64 rows are padding
        nade_zero_rows(d_bias, ld_bias, tracks * Hn + m * D, D, blockIdx.x * 64, 64, N, 512);
        if (nll != nullptr && threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) nll[(size_t)m * N + blockIdx.x * 64 + threadIdx.x] = 0.f;
        return;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbase = blockIdx.x * 64 + w * FWD_R;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const float* __restrict__ wd = w_dec + (size_t)m * D * Hn;
    const int bd_off = tracks * Hn + m * D;

    float a[FWD_R][HQ], h[FWD_R][HQ];
#pragma unroll
    for (int r = 0; r < FWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q)
            a[r][q] = bias[(size_t)min(rbase + r, N - 1) * ld_bias + m * Hn + min(lane + 64 * q, Hn - 1)];     // all in flight together
#pragma unroll
    for (int r = 0; r < FWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = lane + 64 * q, row = rbase + r;
            if (!(row < N && j < Hn)) a[r][q] = 0.f;
            h[r][q] = fast_sigmoid(a[r][q]);
        }
    float u[UT ? FWD_R : 1][UT ? HQ : 1];                // exp(-a), advanced by one multiply per flip (UT)
    float amax = 0.f;                                    // running max of |a| over this lane's states since the last re-derivation
    bool passed40 = false;                               // ... and whether it ever passed the bound (the backward's licence to drop `a`: *unsafe)
    if constexpr (UT) {
#pragma unroll
        for (int r = 0; r < FWD_R; ++r)
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                u[r][q] = fast_exp2(-MNN_LOG2E * a[r][q]);
                amax = fmaxf(amax, fabsf(a[r][q]));
            }
    }
    // lane L owns (row L>>3, visible L&7) of every chunk: it prefetches that v / b_dec and finalises that logit
    const int frow = rbase + (lane >> 3), fi = lane & 7;
    const bool fvalid = frow < N;
    const int frr = fvalid ? frow : N - 1;
    const float rw = (row_weight != nullptr && fvalid) ? row_weight[frow] : 0.f;
    float lp = 0.f;

    WStage<HQ> st;
    st.gload(wd, we, 0, D, Hn);
    st.lstore(wl[0][0], wl[0][1]);
    bool vcur = fvalid && fi < D && vm[(size_t)frr * D + fi] != 0;
    float bcur = fi < D ? bias[(size_t)frr * ld_bias + bd_off + fi] : 0.f;
    __syncthreads();
    const int nch = (D + 7) / 8;
    for (int c = 0; c < nch; ++c) {
        const int i0 = c * 8;
        const int inext = i0 + 8 + fi;
        st.gload(wd, we, i0 + 8, D, Hn);                                      // chunk c+1 (zeros past D)
        // raw, unconditional requests; looked at only at the end of the chunk (evaluated here they were waited for here: s_waitcnt vmcnt(0)
        // behind the weight prefetch just issued, once per chunk)
        const int inc = min(inext, D - 1);
        const uint8_t vraw = vm[(size_t)frr * D + inc];
        const float braw = bias[(size_t)frr * ld_bias + bd_off + inc];
        const unsigned long long mask = __ballot(vcur);                          // bit r*8+ii : v[row r][i0+ii]
        const float* __restrict__ sd = wl[c & 1][0];
        const float* __restrict__ se = wl[c & 1][1];
        float y[8];
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            float wdv[HQ], pr[8];
#pragma unroll
            for (int q = 0; q < HQ; ++q) wdv[q] = sd[ii * W + lane + 64 * q];
            if constexpr (UT) {
                const bool any = (mask & (0x0101010101010101ull << ii)) != 0ull;   // a row of the wave flips at this visible (wave-uniform)
                float wev[HQ], ewv[HQ];
                if (any) {
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        wev[q] = se[ii * W + lane + 64 * q];
                        ewv[q] = fast_exp2(-MNN_LOG2E * wev[q]);                   // ONE exponential per (visible, hidden unit) for all flipping rows
                    }
                }
#pragma unroll
                for (int r = 0; r < FWD_R; ++r) {
                    float acc = h[r][0] * wdv[0];
#pragma unroll
                    for (int q = 1; q < HQ; ++q) acc = fmaf(h[r][q], wdv[q], acc);
                    pr[r] = acc;
                    if ((mask >> (r * 8 + ii)) & 1ull) {                           // wave-uniform: encode v_i = 1 (nade.py:219)
#pragma unroll
                        for (int q = 0; q < HQ; ++q) {
                            a[r][q] += wev[q];                                     // the exact sum (handed to the backward pass; the guard below)
                            u[r][q] *= ewv[q];
                            h[r][q] = fast_rcp(1.0f + u[r][q]);
                            amax = fmaxf(amax, fabsf(a[r][q]));
                        }
                    }
                }
                if (any && __any(amax > 40.0f)) {                                  // rare: re-derive every u from its exact a (see the header)
                    amax = 0.f;
                    passed40 = true;
#pragma unroll
                    for (int r = 0; r < FWD_R; ++r)
#pragma unroll
                        for (int q = 0; q < HQ; ++q) {
                            u[r][q] = fast_exp2(-MNN_LOG2E * a[r][q]);
                            h[r][q] = fast_rcp(1.0f + u[r][q]);
                        }
                }
            } else {
#pragma unroll
                for (int r = 0; r < FWD_R; ++r) {
                    float acc = h[r][0] * wdv[0];
#pragma unroll
                    for (int q = 1; q < HQ; ++q) acc = fmaf(h[r][q], wdv[q], acc);
                    pr[r] = acc;
                    if ((mask >> (r * 8 + ii)) & 1ull) {                             // wave-uniform: encode v_i = 1 (nade.py:219)
#pragma unroll
                        for (int q = 0; q < HQ; ++q) {
                            a[r][q] += se[ii * W + lane + 64 * q];
                            h[r][q] = fast_sigmoid(a[r][q]);
                        }
                    }
                }
            }
            y[ii] = rows8(pr, lane);
        }
        const float tot = vis8(y, lane);
        const int i = i0 + fi;
        if (i < D) {
            const float l = bcur + tot;
            const float pr = fast_sigmoid(l);
            const float qr = fast_sigmoid(-l);      // 1-p without cancellation (closer to the exact value than f32 `1 - p`)
            lp += vcur ? fast_ln(NADE_EPS + pr) : fast_ln(NADE_EPS + qr);
            if (fvalid) {
                if (cond_p != nullptr) cond_p[((size_t)m * N + frow) * D + i] = pr;
                if (d_bias != nullptr) {
                    const float dnll_dp = vcur ? -fast_rcp(NADE_EPS + pr) : fast_rcp(NADE_EPS + qr);
                    d_bias[(size_t)frow * ld_bias + bd_off + i] = rw * dnll_dp * pr * qr;
                }
            }
        }
        st.lstore(wl[(c + 1) & 1][0], wl[(c + 1) & 1][1]);
        vcur = fvalid && inext < D && vraw != 0;
        bcur = inext < D ? braw : 0.f;
        __syncthreads();
    }
    lp += dpp_xor1(lp);
    lp += dpp_xor2(lp);
    lp += swz_xor4(lp);
    if (fi == 0 && fvalid && nll != nullptr) nll[(size_t)m * N + frow] = -lp;
    // hand the final pre-activation a_D (forward-order sum) to the backward kernel
    if (a_final != nullptr) {
#pragma unroll
        for (int r = 0; r < FWD_R; ++r)
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int j = lane + 64 * q, row = rbase + r;
                if (row < N && j < Hn) a_final[((size_t)m * N + row) * Hn + j] = a[r][q];
            }
    }
    // *unsafe counts the waves a pre-activation of which (any row, any hidden unit, any point of the scan) passed |a| = 40; zero (the entry point
    // clears it before the launch) licenses the backward scan, which walks the same values in reverse, to carry exp(-a) instead of a
    // (nade_bwd_kernel<.., USEU>).  Only the multiplicative form tracks the bound: the direct form, and a gated-out launch, count themselves once.
    if (unsafe != nullptr) {
        const bool ok = UT ? !__any(passed40 || amax > 40.0f) : !(blockIdx.x == 0 && blockIdx.y == 0 && w == 0);
        if (!ok && lane == 0) atomicAdd(unsafe, 1);
    }
}

extern "C" int mnn_nade_logprob_fwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                    const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* row_weight,
                                    float* nll, float* cond_p, float* d_bias, float* a_final) {
    return mnn_nade_logprob_fwd_gated(s, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, w_dec, row_weight, nll, cond_p, d_bias, a_final,
                                      nullptr, 0, nullptr, nullptr);
}

extern "C" int mnn_nade_logprob_fwd_gated(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                          const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* row_weight,
                                          float* nll, float* cond_p, float* d_bias, float* a_final, const int* gate, int run_if, const int* n_rows_dev,
                                          int* unsafe) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_logprob_fwd: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(v && bias && w_enc && w_dec, "mnn_nade_logprob_fwd: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_logprob_fwd: ld_bias %d < tracks*(Hn+D)", ld_bias);
    MNN_REQUIRE(d_bias == nullptr || row_weight != nullptr, "mnn_nade_logprob_fwd: d_bias needs row_weight");
    dim3 grid(cdiv(N, 64), tracks);
    hipStream_t st = (hipStream_t)s;
    // the density-gated DENSE launch of the 16-bit modes (gate given, run_if = 1) advances the hidden states multiplicatively (UT); every other
    // caller -- the f32 parity mode, conditionals on demand, Hn <= 128 -- keeps the direct sigmoid
#define FWD(HQ, UT) hipLaunchKernelGGL((nade_fwd_kernel<HQ, UT>), grid, dim3(512), 0, st, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, \
                                       w_dec, row_weight, nll, cond_p, d_bias, a_final, gate, run_if, n_rows_dev, unsafe)
    if (unsafe != nullptr) MNN_HIP(mnn_zero_async(unsafe, sizeof(int), sizeof(int), 1, st));      // counted by the launch below (a fill KERNEL: common.h on memset nodes)
    static const bool no_ut = getenv("MNN_NADE_FWD_NO_UT") != nullptr;          // (tests: the direct form as the comparison partner)
    if (Hn <= 64) FWD(1, false);
    else if (Hn <= 128) FWD(2, false);
    else if (gate != nullptr && run_if == 1 && !no_ut) FWD(4, true);
    else FWD(4, false);
#undef FWD
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// backward: reverse scan over the visible order
// ----------------------------------------------------------------------------------------------
// Staging for the backward kernel.  HQ == 4: lane-major in LDS (the hidden units l, l+64, l+128, l+192 side by side), so the scan reads
// the four values a lane needs with one 16-byte load per matrix.  Thread (visible t >> 6, slot l = t & 63) fetches exactly those four
// values (four loads, each 256 contiguous bytes per wave) and stores them with ONE conflict-free 16-byte LDS store per matrix.
// HQ consecutive floats as one LDS access (HQ = 2: 8 bytes, HQ = 4: 16 bytes)
template <int HQ> struct LaneVec;
template <> struct LaneVec<2> { typedef float2 T; };
template <> struct LaneVec<4> { typedef float4 T; };
template <int HQ>
__device__ __forceinline__ void lv_store(float* p, const float (&x)[HQ]) {
    if constexpr (HQ == 4) *reinterpret_cast<float4*>(p) = make_float4(x[0], x[1], x[2], x[3]);
    else *reinterpret_cast<float2*>(p) = make_float2(x[0], x[1]);
}
template <int HQ>
__device__ __forceinline__ void lv_load(const float* p, float (&x)[HQ]) {
    if constexpr (HQ == 4) { const float4 t = *reinterpret_cast<const float4*>(p); x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w; }
    else { const float2 t = *reinterpret_cast<const float2*>(p); x[0] = t.x; x[1] = t.y; }
}
template <int HQ>
__device__ __forceinline__ void bwd_gload(WStage<HQ>& st, const float* __restrict__ wd, const float* __restrict__ we, int i0, int D, int Hn, int ld) {
    if constexpr (HQ >= 2) {
        const int i = i0 + (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;     // the visible is wave-uniform
        // UNCONDITIONAL loads from clamped addresses (rows outside [0, D) and hidden units past the slice are never used: their visibles are
        // skipped, their lanes never stored): a predicated load is a branch around the load, and the loads of a chunk then drain one by one
        const int ic = min(max(i, 0), D - 1);                                                                // scalar
        const float* __restrict__ pd = wd + (size_t)ic * ld;
        const float* __restrict__ pe = we + (size_t)ic * ld;
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int col = min(l + 64 * q, Hn - 1);
            st.rd[q] = pd[col];
            st.re[q] = pe[col];
        }
    } else {
        st.gload(wd, we, i0, D, Hn, ld);
    }
}
template <int HQ>
__device__ __forceinline__ void bwd_lstore(const WStage<HQ>& st, float* __restrict__ sd, float* __restrict__ se) {
    if constexpr (HQ >= 2) {
        const int pos = (int)(threadIdx.x >> 6) * (64 * HQ) + HQ * (int)(threadIdx.x & 63);
        lv_store<HQ>(sd + pos, st.rd);
        lv_store<HQ>(se + pos, st.re);
    } else {
        st.lstore(sd, se);
    }
}

// RG = groups of 8 rows per wave (rows per workgroup: 64 RG).  RG = 2 halves the f32 atomics per row (one add per visible and hidden unit
// per 128 rows) and the LDS reads / exchanges per FMA, at twice the state registers (HQ = 2: 4 x 16 x 2 = 128): two waves per SIMD.
// USEU: the multiplicative form of the state updates for DENSE batches (see `useu` below); chosen per LAUNCH by a device word (both instantiations
// are launched, one leaves): as a per-wave choice inside one kernel the two flip paths cost 175 registers (one workgroup per CU: 3.1 -> 5.6 ms).
template <int HQ, int RG, bool USEU>
__global__ void __launch_bounds__(512)
nade_bwd_kernel(int tracks, int N, int D, int HnT, int nslice, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias,
                int ld_bias, const float* __restrict__ w_enc, const float* __restrict__ w_dec, const float* __restrict__ a_final,
                float* __restrict__ d_bias, float* __restrict__ d_w_enc, float* __restrict__ d_w_dec, const int* __restrict__ n_rows_dev,
                const int* __restrict__ unsafe) {
    constexpr int W = HQ * 64;
    __shared__ __attribute__((aligned(16))) float wl[2][2][8 * W];
    if (unsafe != nullptr && (*unsafe == 0) != USEU) return;                // the pair of launches of mnn_nade_logprob_bwd: uniform exit
    // [buffer][wave][visible-in-half-chunk][d w_dec | d w_enc][hidden]: one exchange per 4 visibles.  Two buffers where they fit beside a second
    // workgroup of the CU (HQ <= 2: 2 x 32 KB + 16 KB of weights = 80 KB): an exchange then needs ONE barrier (stores -> barrier -> sums; the next
    // exchange stores into the other buffer, and a wave gets there only through the barrier every wave reaches after its previous sums), and
    // the barrier at the end of a chunk goes too -- 2 barriers per 8 visibles instead of 5: 3.60 -> 3.38 ms at [1024,256,88,5].
    // (Measured and dropped: the 64 v_readlane per chunk that broadcast d nll / d logit replaced by scalar loads of the same values --
    // s_load_dwordx2 per row and visible pair; the 16 + 16 extra scalar registers spill, 4.27 ms.)
    constexpr int NRED = HQ <= 2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float red_[NRED][8][4][2][W];
    // blockIdx.y = track * nslice + hidden slice: the backward scan is separable over hidden units (only the forward logit sums over them), so a
    // wide layer may run as nslice narrower workgroups (fewer registers and less LDS each: more of them resident per CU)
    const int m = blockIdx.y / nslice, hb = (blockIdx.y - m * nslice) * W;
    const int Hn = min(W, HnT - hb);                                           // hidden units of this slice
    if (nade_rows_beyond(n_rows_dev, blockIdx.x * (64 * RG))) {                // compacted ragged batch: padding rows only -- d b_enc = 0
        nade_zero_rows(d_bias, ld_bias, m * HnT + hb, Hn, blockIdx.x * (64 * RG), 64 * RG, N, 512);
        return;
    }
    const int Nv = n_rows_dev != nullptr ? min(*n_rows_dev, N) : N;           // rows with a forward result
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BWD_R = 8 * RG;                                              // rows per wave
    const int rbase = blockIdx.x * (64 * RG) + w * BWD_R;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * HnT + hb;
    const float* __restrict__ wd = w_dec + (size_t)m * D * HnT + hb;
    const int dl_off = tracks * HnT + m * D;

    // Sparsity (exact): `a` only changes at visibles with v = 1 (nade.py:219), so h = sigmoid(a) is cached
    // and recomputed only there, and sum_i dl_i * w_dec[i] is accumulated per constant-h segment (c) and
    // folded into G with ONE h(1-h) factor when the segment ends.  a_D comes from the forward kernel.
    // USEU (dense batches whose every row the forward's multiplicative form vouched for -- |a| <= 40 throughout: *unsafe == 0): the state
    // register of a (row, unit) carries u = exp(-a) instead of a -- a flip is u *= exp(+w_enc[i]) (one exponential per (visible, unit) for all the
    // rows that flip there), h = 1 / (1 + u): mul + add + rcp instead of sub + mul + exp + add + rcp.  The scan revisits the forward's
    // pre-activations in reverse, so u stays in f32 range and nothing needs `a` itself.
    constexpr bool useu = USEU;
    float a[BWD_R][HQ], h[BWD_R][HQ], c[BWD_R][HQ], G[BWD_R][HQ];
    // all a_D loads in flight together (unconditional, clamped row / hidden unit), masked afterwards: behind `valid ? load : 0` each load
    // waited for its own round trip (s_waitcnt vmcnt(0) per element at the start of every workgroup)
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q)
            a[r][q] = a_final[((size_t)m * N + min(rbase + r, N - 1)) * HnT + hb + min(lane + 64 * q, Hn - 1)];
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = lane + 64 * q, row = rbase + r;
            if (!(row < Nv && j < Hn)) a[r][q] = 0.f;           // (rows behind a compacted batch's valid ones: a_final was never written for them)
            h[r][q] = fast_sigmoid(a[r][q]);
            if constexpr (useu) a[r][q] = fast_exp2(-MNN_LOG2E * a[r][q]);   // the register now carries u = exp(-a)
            G[r][q] = 0.f;
            c[r][q] = 0.f;
        }
    const int fi = lane & 7;
    int frr[RG];
    bool fvalid[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int frow = rbase + 8 * g + (lane >> 3);
        fvalid[g] = frow < N;
        frr[g] = fvalid[g] ? frow : N - 1;
    }
    const int nch = (D + 7) / 8;
    WStage<HQ> st;
    bwd_gload<HQ>(st, wd, we, (nch - 1) * 8, D, Hn, HnT);
    bwd_lstore<HQ>(st, wl[0][0], wl[0][1]);
    const int icur = (nch - 1) * 8 + fi;
    bool vcur[RG];
    float dcur[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        vcur[g] = fvalid[g] && icur < D && vm[(size_t)frr[g] * D + icur] != 0;
        dcur[g] = (fvalid[g] && icur < D) ? d_bias[(size_t)frr[g] * ld_bias + dl_off + icur] : 0.f;
    }
    __syncthreads();
    for (int cc = 0; cc < nch; ++cc) {
        const int i0 = (nch - 1 - cc) * 8;
        const int inext = i0 - 8 + fi;
        bwd_gload<HQ>(st, wd, we, i0 - 8, D, Hn, HnT);                             // next (lower) chunk, zeros below 0
        // the next chunk's v byte and d nll / d logit: RAW, unconditional loads (clamped index); they are looked at -- compared, masked -- only
        // at the END of this chunk.  Evaluated here (`!= 0`, `valid ? x : 0`), each load was followed by s_waitcnt vmcnt(0): one full memory
        // round trip per chunk, the weight prefetch just issued included
        uint8_t vraw[RG];
        float draw[RG];
        unsigned long long mask[RG], many = 0ull;
        const int inc = max(inext, 0);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            vraw[g] = vm[(size_t)frr[g] * D + inc];
            draw[g] = d_bias[(size_t)frr[g] * ld_bias + dl_off + inc];
            mask[g] = __ballot(vcur[g]);
            many |= mask[g];
        }
        const float* __restrict__ sd = wl[cc & 1][0];
        const float* __restrict__ se = wl[cc & 1][1];
#pragma unroll
        for (int half = 1; half >= 0; --half) {
            float accd[4][HQ], acce[4][HQ];
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int ii = half * 4 + k;
                const int i = i0 + ii;
#pragma unroll
                for (int q = 0; q < HQ; ++q) { accd[k][q] = 0.f; acce[k][q] = 0.f; }
                if (i >= D) continue;                                           // block-uniform (tail chunk)
                float wev[HQ], wdv[HQ];
                if constexpr (HQ >= 2) {                                        // lane-major staging: one 8/16-byte LDS load per matrix
                    lv_load<HQ>(sd + ii * W + HQ * lane, wdv);
                    lv_load<HQ>(se + ii * W + HQ * lane, wev);
                } else {
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        wdv[q] = sd[ii * W + lane + 64 * q];
                        wev[q] = se[ii * W + lane + 64 * q];
                    }
                }
                // the rows are independent: first the (rare) state changes of the rows with v_i = 1 -- skipped with ONE scalar test when
                // none of the wave's 8 rows has one (4 visibles in 5 at rho = 0.03) --, then the FMAs of all 8 rows, straight-line
                if ((many & (0x0101010101010101ull << ii)) != 0ull) {
                    if constexpr (useu) {
                        float ewv[HQ];
#pragma unroll
                        for (int q = 0; q < HQ; ++q) ewv[q] = fast_exp2(MNN_LOG2E * wev[q]);          // exp(+w_enc[i]): once for every row that flips here
#pragma unroll
                        for (int r = 0; r < BWD_R; ++r) {
                            if ((mask[r >> 3] >> ((r & 7) * 8 + ii)) & 1ull) {
#pragma unroll
                                for (int q = 0; q < HQ; ++q) {
                                    G[r][q] = fmaf(c[r][q], fmaf(-h[r][q], h[r][q], h[r][q]), G[r][q]);
                                    c[r][q] = 0.f;
                                    acce[k][q] += G[r][q];
                                    a[r][q] *= ewv[q];           // u_i = u_{i+1} exp(+w_enc[i])
                                    h[r][q] = fast_rcp(1.0f + a[r][q]);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < BWD_R; ++r) {
                            if ((mask[r >> 3] >> ((r & 7) * 8 + ii)) & 1ull) {
#pragma unroll
                                for (int q = 0; q < HQ; ++q) {
                                    G[r][q] = fmaf(c[r][q], fmaf(-h[r][q], h[r][q], h[r][q]), G[r][q]);   // close the segment that used a_{i+1}
                                    c[r][q] = 0.f;
                                    acce[k][q] += G[r][q];       // d w_enc[i] += v_i * G_{i+1}
                                    a[r][q] -= wev[q];           // a_i = a_{i+1} - v_i * w_enc[i]
                                    h[r][q] = fast_sigmoid(a[r][q]);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < BWD_R; ++r) {
                    const float dl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dcur[r >> 3]), (r & 7) * 8 + ii));
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        accd[k][q] = fmaf(dl, h[r][q], accd[k][q]);
                        c[r][q] = fmaf(dl, wdv[q], c[r][q]);
                    }
                }
            }
            // The next chunk's staged weights go to LDS HERE (first half only): their global loads were issued at the top of the chunk
            // and the only younger memory operations of this wave are none -- after the exchanges the wait for them would also sit
            // behind this chunk's f32 atomics.  The other buffer of wl was last read in the previous chunk (barrier at its end).
            if (half == 1) bwd_lstore<HQ>(st, wl[(cc + 1) & 1][0], wl[(cc + 1) & 1][1]);
            // one cross-wave exchange per 4 visibles (the per-visible barrier was 60 % of the wave time)
            float (*red)[4][2][W] = red_[(2 * cc + 1 - half) & (NRED - 1)];
            if (NRED == 1) __syncthreads();              // the previous exchange has been read
            if constexpr (HQ >= 2) {
                // exchange slots are LANE-major (the four hidden units lane, lane+64, lane+128, lane+192 of a lane side by side): one
                // 16-byte LDS store per (visible, matrix) instead of four 4-byte ones, and the summing thread -- one per (visible,
                // matrix, lane) -- reads 8 x 16 bytes instead of 32 x 4; its four atomics each still cover 256 contiguous bytes per wave
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lv_store<HQ>(&red[w][k][0][HQ * lane], accd[k]);
                    lv_store<HQ>(&red[w][k][1][HQ * lane], acce[k]);
                }
                __syncthreads();
                {
                    const int k = threadIdx.x >> 7, which = (threadIdx.x >> 6) & 1;       // 4 visibles x 2 matrices x 64 lanes = 512 threads
                    const int i = i0 + half * 4 + k;
                    float sum[HQ];
#pragma unroll
                    for (int q = 0; q < HQ; ++q) sum[q] = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 8; ++ww) {
                        float pq[HQ];
                        lv_load<HQ>(&red[ww][k][which][HQ * lane], pq);
#pragma unroll
                        for (int q = 0; q < HQ; ++q) sum[q] += pq[q];
                    }
                    if (i < D) {
                        // one f32 atomic per (visible, hidden unit) and 64-row workgroup, 256 contiguous bytes per wave.  (Per-workgroup slabs
                        // + a reduction pass instead -- bit-reproducible sums -- were measured slower, 5.1 vs 4.0 ms at [1024,256,88,5], and
                        // removed in round 4: the atomics ride under the scan's VALU work, the slab stores and their 3.7 GB read-back do not.)
                        float* dst = (which == 0 ? d_w_dec : d_w_enc) + ((size_t)m * D + i) * HnT + hb + lane;
#ifdef NADE_BWD_NO_ATOMIC       // development only (timing the scan without its f32 atomics: results are wrong)
                        if (sum[0] == 1.2345e30f) dst[0] = sum[0];
#else
#pragma unroll
                        for (int q = 0; q < HQ; ++q)
                            if (lane + 64 * q < Hn) atomicAdd(dst + 64 * q, sum[q]);
#endif
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        red[w][k][0][lane + 64 * q] = accd[k][q];
                        red[w][k][1][lane + 64 * q] = acce[k][q];
                    }
                __syncthreads();
                for (int e = threadIdx.x; e < 4 * 2 * W; e += 512) {
                    const int k = e / (2 * W), which = (e / W) & 1, j = e % W;
                    const int i = i0 + half * 4 + k;
                    if (j < Hn && i < D) {
                        float sum = 0.f;
#pragma unroll
                        for (int ww = 0; ww < 8; ++ww) sum += red[ww][k][which][j];
                        atomicAdd((which == 0 ? d_w_dec : d_w_enc) + ((size_t)m * D + i) * HnT + hb + j, sum);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
            const bool ok = fvalid[g] && inext >= 0;
            vcur[g] = ok && vraw[g] != 0;
            dcur[g] = ok ? draw[g] : 0.f;
        }
        if (NRED == 1) __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) G[r][q] = fmaf(c[r][q], fmaf(-h[r][q], h[r][q], h[r][q]), G[r][q]);
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = lane + 64 * q, row = rbase + r;
            if (row < N && j < Hn) __builtin_nontemporal_store(G[r][q], &d_bias[(size_t)row * ld_bias + m * HnT + hb + j]);      // written once, read by a later kernel
        }
}

extern "C" int mnn_nade_logprob_bwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                    const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* a_final,
                                    float* d_bias, float* d_w_enc, float* d_w_dec, const int* n_rows_dev, const int* unsafe) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_logprob_bwd: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(v && bias && w_enc && w_dec && a_final && d_bias && d_w_enc && d_w_dec, "mnn_nade_logprob_bwd: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_logprob_bwd: ld_bias too small");
    hipStream_t st = (hipStream_t)s;
    // The backward scan is separable over hidden units, so a wide layer runs as 128-wide slices (HQ = 2: 116 VGPRs, 80 KB of LDS -> two
    // workgroups per CU).  Measured and removed in round 4 (never the default): one 256-wide workgroup (HQ = 4), 64-wide slices, and 128 rows
    // per workgroup (two row groups per wave: 4.98 vs 3.94 ms at [1024,256,88,5] -- 187 registers leave two waves per SIMD instead of four).
    // unsafe (the counter the forward's density-gated dense launch left: mnn_nade_logprob_fwd_gated): 0 -> the multiplicative instantiation;
    // both are launched and the one the word does not pick leaves at its first instruction (NULL: the direct form only)
#define BWD(HQ, NS, U, S) hipLaunchKernelGGL((nade_bwd_kernel<HQ, 1, U>), dim3(cdiv(N, 64), tracks * (NS)), dim3(512), 0, st, tracks, N, D, Hn, NS, v, \
                                             v_track_stride, bias, ld_bias, w_enc, w_dec, a_final, d_bias, d_w_enc, d_w_dec, n_rows_dev, S)
    if (Hn <= 64) BWD(1, 1, false, nullptr);
    else {
        if (unsafe != nullptr) BWD(2, cdiv(Hn, 128), true, unsafe);
        BWD(2, cdiv(Hn, 128), false, unsafe);
    }
#undef BWD
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// sampling: one wave per (row, track); deterministic order (DESIGN.md "Deterministic sampling"):
//   lane l owns hidden j = l + 64 q, q = 0..3 ; acc_l = fma-chain over q ; xor-butterfly 32..1 ;
//   logit = b_dec + acc ; p = det_sigmoid(logit) ; draw = u < det_sigmoid(logit / T)
// The scan is a serial chain of D conditionals per row, so everything that does not depend on the previous draw is kept off it:
//   * the hidden state h = det_sigmoid(a) is cached and recomputed only after a draw of 1 (the same value otherwise);
//   * the two weight rows and b_dec of a visible are fetched EIGHT visibles ahead into a register ring (a fetch per visible on the
//     chain was an L2 round trip per conditional);
//   * the uniforms: lane l evaluates the Philox block (first block of the chunk + l) once per 256 elements, the chain reads its
//     word with one v_readlane (every lane used to run the ten rounds for every visible);
//   * the xor butterfly runs on permlane swaps / DPP (each step adds the same two numbers as the shuffle form: same bits);
//   * probabilities and draws are parked in LDS; the log terms are evaluated 64 at a time after the scan and added in visible order
//     (the same sum as adding inside the loop), samples leave in one strided pass.
// 487 -> see profiles/round1_d_notes.md (us per call at D = 440, Hn = 256).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_xor_sum(float x) {
    {
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);     // x[l] + x[l ^ 32]
        x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);     // x[l] + x[l ^ 16]
        x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    x = x + dpp_xor8(x);
    x = x + dpp_xor4(x);
    x = x + dpp_xor2(x);
    x = x + dpp_xor1(x);
    return x;
}

// The draw 1[u < det_sigmoid(x)] (or det_sigmoid(x) >= 1/2) decided WITHOUT the 35-operation deterministic sigmoid on the serial chain:
// the hardware exp2 / rcp give the probability to ~1e-5 relative (argument scaling of exp2 included), which settles every draw
// whose uniform is not within 1e-4 (relative) of it; the rare rest takes the exact comparison.  The outcome is the exact one always.
__device__ __forceinline__ float sig_approx(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-x * 1.4426950408889634f));
}
__device__ __forceinline__ bool draw_below(float u, float x) {          // u < det_sigmoid(x); wave-uniform operands
    const float r = sig_approx(x);
    const float d = u - r;
    if (__builtin_amdgcn_ballot_w64(fabsf(d) > 1e-4f * r + 1e-30f) != 0ull) return __builtin_amdgcn_ballot_w64(d < 0.f) != 0ull;
    return __builtin_amdgcn_ballot_w64(u < det_sigmoid(x)) != 0ull;
}
__device__ __forceinline__ bool prob_at_least_half(float x) {           // det_sigmoid(x) >= 0.5f
    const float d = sig_approx(x) - 0.5f;
    if (__builtin_amdgcn_ballot_w64(fabsf(d) > 1e-4f) != 0ull) return __builtin_amdgcn_ballot_w64(d > 0.f) != 0ull;
    return __builtin_amdgcn_ballot_w64(det_sigmoid(x) >= 0.5f) != 0ull;
}

// one (generator, track) of a launch: where its biases, weights, uniforms and outputs are
struct SampleJob {
    const float* bias; int ld_bias; int enc_off, dec_off;    // b_enc at bias[row, enc_off ..], b_dec at bias[row, dec_off ..]
    const float* w_enc; const float* w_dec;                  // [D, Hn] of this track
    uint64_t seed; uint32_t elem0;                           // Philox key; element index of visible 0 (m D inside a MultiNADE, 0 otherwise)
    uint8_t* samples; float* nll;                            // samples[row * s_row_stride + i * s_elem_stride]; nll [N] or NULL
};
#define SAMPLE_MAX_JOBS 8
struct SampleJobs { SampleJob job[SAMPLE_MAX_JOBS]; };

// TMODE: 0 = threshold draws (temperature None / <= 0), 1 = temperature 1, 2 = any other temperature.  FULL: Hn == 256, no lane is idle.
template <int TMODE, bool FULL, bool SPEC>
__global__ void __launch_bounds__(256)
nade_sample_kernel(SampleJobs J, int N, int D, int Hn, float temperature, uint32_t row0, uint32_t sub, long s_row_stride, int s_elem_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nade_sample_smem[];       // per wave: logit / log term [Dp] f32, b_dec [Dp] f32, draws [Dp] u8, 256 uniforms
    // blockIdx.y = job: a track of one MultiNADE (mnn_nade_sample) or one of several generators sampled together (mnn_nade_sample_multi)
    const SampleJob& jb = J.job[blockIdx.y];
    const float* __restrict__ bias = jb.bias;
    const int ld_bias = jb.ld_bias;
    const uint64_t seed = jb.seed;
    uint8_t* __restrict__ samples = jb.samples;
    float* __restrict__ nll = jb.nll;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    if (row >= N) return;                                   // wave-uniform; the kernel has no workgroup barrier
    const int Dp = (D + 3) & ~3;
    float* sp = reinterpret_cast<float*>(nade_sample_smem) + (size_t)wv * Dp;
    float* sbd = reinterpret_cast<float*>(nade_sample_smem) + (size_t)(4 + wv) * Dp;
    unsigned char* son = nade_sample_smem + (size_t)32 * Dp + (size_t)wv * Dp;
    float* su = reinterpret_cast<float*>(nade_sample_smem + (size_t)36 * Dp) + wv * 256;    // uniforms of Philox blocks b0 .. b0 + 63
    const float* __restrict__ we = jb.w_enc;
    const float* __restrict__ wd = jb.w_dec;
    const float* __restrict__ bd = bias + (size_t)row * ld_bias + jb.dec_off;
    float a[4], h[4];
    bool in[4];
    int off[4];                                             // hidden index of (lane, q), 0 where there is none: loads are never predicated
#pragma unroll                                              // (a load under a branch is waited for right behind it -- the ring would be no ring)
    for (int q = 0; q < 4; ++q) {
        in[q] = FULL || lane + 64 * q < Hn;
        off[q] = in[q] ? lane + 64 * q : 0;
        const float av = bias[(size_t)row * ld_bias + jb.enc_off + off[q]];
        a[q] = in[q] ? av : 0.f;
        h[q] = det_sigmoid(a[q]);
    }
    constexpr int RING = 8;                                 // visibles in flight: RING x (time per visible) must cover an L2 round trip
    float wdr[RING][4], wer[RING][4];                       // ring: the weight rows of visibles i .. i + RING - 1
    auto fetch = [&](int k, int i) {
        const int ii = min(i, D - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wdr[k][q] = wd[(size_t)ii * Hn + off[q]];
            wer[k][q] = we[(size_t)ii * Hn + off[q]];
        }
    };
    for (int i = lane; i < D; i += 64) sbd[i] = bd[i];      // the row's b_dec: one coalesced pass into LDS, a broadcast read per visible
#pragma unroll
    for (int k = 0; k < RING; ++k) {
        fetch(k, k);
        __builtin_amdgcn_sched_barrier(0);                  // issue order = ring order, in the prologue as in the loop (the waits count loads)
    }
    const uint32_t e0 = jb.elem0;                           // element index of visible 0 (RNG contract: elem = m D + i)
    uint32_t b0 = e0 >> 2;                                  // lane l holds the uniforms of Philox block b0 + l
    auto refill = [&]() {                                   // lane l: the four uniforms of Philox block b0 + l, parked in LDS (one broadcast read per visible)
        float u4[4];
        philox_uniform4(seed, MNN_STREAM_NADE, row0 + (uint32_t)row, sub, b0 + (uint32_t)lane, u4);
        *reinterpret_cast<float4*>(su + 4 * lane) = make_float4(u4[0], u4[1], u4[2], u4[3]);
    };
    if (TMODE != 0) refill();
    auto dot = [&](int k) {                                 // sum_j h_j w_dec[visible of ring slot k][j], the contract's order
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc = fmaf(h[q], in[q] ? wdr[k][q] : 0.f, acc);
        return wave_xor_sum(acc);
    };
    // the uniform of visible i, fetched one visible ahead (with the rare Philox refill) so that the block below -- speculative dot
    // product next to the draw -- stays one straight line the scheduler can interleave
    auto uniform_of = [&](int i) {
        const uint32_t e = e0 + (uint32_t)min(i, D - 1);
        if ((e >> 2) >= b0 + 64u) {                         // uniform: next 64 Philox blocks (the previous ones have all been read)
            b0 += 64u;
            refill();
        }
        return su[e - 4u * b0];
    };
    float u_cur = TMODE != 0 ? uniform_of(0) : 0.f;
    float acc = dot(0);
    for (int i0 = 0; i0 < D; i0 += RING) {
#pragma unroll
        for (int k = 0; k < RING; ++k) {
            const int i = i0 + k;
            if (i < D) {                                    // uniform
                // speculation: if this draw is 0 the hidden state does not change, and visible i + 1's dot product is this one -- it has
                // no dependence on the sigmoid / draw chain below and fills its issue slots (a draw of 1 recomputes it)
                const float spec = SPEC ? dot((k + 1) % RING) : 0.f;
                const float l = sbd[i] + acc;
                bool on;
                if (TMODE != 0) {
                    on = draw_below(u_cur, TMODE == 1 ? l : l / temperature);
                } else {
                    on = prob_at_least_half(l);              // nade.py:278-279
                }
                if (on) {                                   // uniform (every lane holds the same p and u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a[q] = a[q] + (in[q] ? wer[k][q] : 0.f);
                        h[q] = det_sigmoid(a[q]);
                    }
                    acc = dot((k + 1) % RING);
                } else {
                    acc = SPEC ? spec : dot((k + 1) % RING);
                }
                sp[i] = l;                                  // every lane holds the same logit and draw: one merged LDS write each;
                son[i] = on ? 1 : 0;                        // p = det_sigmoid(logit) is evaluated after the scan, 64 visibles at a time
                if (TMODE != 0) u_cur = uniform_of(i + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(k, i + RING);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // after the scan: samples out, log terms 64 at a time, then their sum in visible order (LDS operations of one wave execute in order)
    for (int i = lane; i < D; i += 64) {
        const float p = det_sigmoid(sp[i]);
        const bool on = son[i] != 0;
        samples[(size_t)row * s_row_stride + (size_t)i * s_elem_stride] = on ? 1 : 0;
        sp[i] = on ? logf(NADE_EPS + p) : logf(NADE_EPS + (1.0f - p));
    }
    if (nll != nullptr) {
        float logp = 0.f;
        for (int i = 0; i < D; ++i) logp += sp[i];
        if (lane == 0) nll[row] = -logp;
    }
}

// ----------------------------------------------------------------------------------------------
// The same scan G = 8 or 16 visibles at a time.  Between two draws of 1 the hidden state does not move, so the logits of the next visibles are
// all dot products with the SAME h: a pass evaluates the G logits of an aligned chunk at once -- 4 G FMAs (packed in pairs), one reduce-scatter
// of the xor butterfly (every level adds the same two numbers as wave_xor_sum: same bits; lane l ends with the sum of visible l >> 3, or l >> 2),
// G sigmoids and comparisons in G lane groups -- and the ballot names the first draw of 1 (everything before it is a settled 0).  That one
// flips the state and the chunk is re-evaluated from the visible behind it.  Passes per row: D / G + (number of ones) instead of D dependent
// conditionals (piano-roll rows: ~13 ones in 440).  One wave per workgroup; the weight rows of a chunk arrive by LDS DMA (one
// 16-byte-per-lane copy per row: a ring of chunks in flight, counted waits) -- a register ring that deep would need more than vmcnt's 63 loads
// in flight.  A lone wave spends ~10 cycles per instruction of this dependent chain, so the wider pass costs little more than the narrow one:
// G = 16 while every row has a CU to itself (96 KB of ring), G = 8 (64 KB: two workgroups per CU) for larger batches.
// Needs Hn % 4 == 0 and 16-byte aligned weight matrices (mnn_nade_sample falls back to nade_sample_kernel otherwise).
// profiles/tools/sample_chunk_trace.py: the stage clocks.  [72 rows, D = 440, Hn = 256, piano-roll-like draws: 112 -> see profiles/round6_e_sampling_chunks.md]
// ----------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* nade_gas_ptr_t;
typedef __attribute__((address_space(3))) void* nade_lds_ptr_t;
typedef float nade_f32x2 __attribute__((ext_vector_type(2)));
constexpr int sch_nb(int g) { return g == 16 ? 3 : 4; }      // ring depth: 3 x 32 KB (G = 16) or 4 x 16 KB (G = 8)
#ifdef SCH_TRACE        // development only (profiles/tools/sample_chunk_trace.py): per chunk of row 0 / job 0, [wall clock at the wait | after it | after the passes | shader clock there]
__device__ long long sch_trace[8][256];
extern "C" int mnn_sch_trace_read(long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(sch_trace), sizeof(sch_trace)) == hipSuccess ? 0 : 1; }
#define SCH_TR(k, c) do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (c) < 256) sch_trace[k][c] = (k) == 3 ? clock64() : wall_clock64(); } while (0)
#else
#define SCH_TR(k, c) do { } while (0)
#endif
template <int TMODE, bool FULL, int G>
__global__ void __launch_bounds__(64)
nade_sample_chunk_kernel(SampleJobs J, int N, int D, int Hn, float temperature, uint32_t row0, uint32_t sub, long s_row_stride, int s_elem_stride) {
    constexpr int NB = sch_nb(G);
    constexpr int KSH = G == 16 ? 2 : 3;                      // lane l decides visible l >> KSH of the chunk
    constexpr int WAITN = (NB - 2) * 2 * G;                   // copies that may still be in flight when a chunk is needed: the NB - 2 chunks behind it
    // ring [NB][w_dec | w_enc][G][256] f32 (64 KB) | logit / log term [Dp] f32 | b_dec [Dp] f32 | 256 uniforms | draws [Dp] u8
    extern __shared__ __attribute__((aligned(16))) unsigned char nade_sample_smem[];
    float* ring = reinterpret_cast<float*>(nade_sample_smem);
    const int Dp = (D + 3) & ~3;
    float* sp = ring + NB * 2 * G * 256;
    float* sbd = sp + Dp;
    float* su = sbd + Dp;
    unsigned char* son = reinterpret_cast<unsigned char*>(su + 256);
    const SampleJob& jb = J.job[blockIdx.y];
    const float* __restrict__ bias = jb.bias;
    const int ld_bias = jb.ld_bias;
    const uint64_t seed = jb.seed;
    uint8_t* __restrict__ samples = jb.samples;
    float* __restrict__ nll = jb.nll;
    const int lane = threadIdx.x;
    const int row = blockIdx.x;                              // < N by the grid
    const float* __restrict__ we = jb.w_enc;
    const float* __restrict__ wd = jb.w_dec;
    const float* __restrict__ bd = bias + (size_t)row * ld_bias + jb.dec_off;
    // LDS DMA of chunk c into ring slot b: lane l copies floats 4l .. 4l + 3 of a row (inside the row for narrow layers: what lands past Hn is never read)
    const int src_off = min(4 * lane, Hn - 4);
    auto stage = [&](int b, int c) {
        if (FULL && D >= G) {
            // Hn == 256: the rows of a chunk are 1 KB apart in memory AND in the ring, so the instruction offset (added to both addresses) walks
            // them -- four copies per address / M0 set-up instead of one (a set-up is ~10 scalar + vector instructions: 320 -> ~100 ns per chunk).
            // Chunks past the end (issued to keep the wait count constant, never read) and the ragged last one start at row D - G: every copy
            // stays inside the matrix; the last chunk's rows are then read from where they landed (`rsh` below).
            const int r0 = min(c * G, D - G);
#pragma unroll
            for (int k4 = 0; k4 < G; k4 += 4) {
                const float* gd = wd + (size_t)(r0 + k4) * 256 + 4 * lane;
                const float* ge = we + (size_t)(r0 + k4) * 256 + 4 * lane;
                float* ld_ = ring + ((b * 2 + 0) * G + k4) * 256;
                float* le_ = ring + ((b * 2 + 1) * G + k4) * 256;
#define SCH_COPY4(G_, L_) __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(G_), (nade_lds_ptr_t)(L_), 16, 0, 0);    __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(G_), (nade_lds_ptr_t)(L_), 16, 1024, 0); \
                          __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(G_), (nade_lds_ptr_t)(L_), 16, 2048, 0); __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(G_), (nade_lds_ptr_t)(L_), 16, 3072, 0)
                SCH_COPY4(gd, ld_);
                SCH_COPY4(ge, le_);
#undef SCH_COPY4
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int ii = min(c * G + k, D - 1);
            __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(wd + (size_t)ii * Hn + src_off), (nade_lds_ptr_t)(ring + ((b * 2 + 0) * G + k) * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((nade_gas_ptr_t)(we + (size_t)ii * Hn + src_off), (nade_lds_ptr_t)(ring + ((b * 2 + 1) * G + k) * 256), 16, 0, 0);
        }
    };
    float a[4], h[4];
    bool in[4];
    int off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        in[q] = FULL || lane + 64 * q < Hn;
        off[q] = in[q] ? lane + 64 * q : 0;
        const float av = bias[(size_t)row * ld_bias + jb.enc_off + off[q]];
        a[q] = in[q] ? av : 0.f;
        h[q] = det_sigmoid(a[q]);
    }
    for (int i = lane; i < D; i += 64) sbd[i] = bd[i];
    const uint32_t e0 = jb.elem0;
    uint32_t b0 = e0 >> 2;
    auto refill = [&]() {                                    // (an LDS write waits for every copy in flight -- the compiler cannot tell them apart: once per ~31 chunks)
        float u4[4];
        philox_uniform4(seed, MNN_STREAM_NADE, row0 + (uint32_t)row, sub, b0 + (uint32_t)lane, u4);
        *reinterpret_cast<float4*>(su + 4 * lane) = make_float4(u4[0], u4[1], u4[2], u4[3]);
    };
    if (TMODE != 0) refill();
    asm volatile("" ::"v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]));     // the initial states are evaluated HERE: behind the copies they would wait for all 64 of them
    __builtin_amdgcn_sched_barrier(0);                       // the loads above are older than every DMA: the counted waits below cover them
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) stage(b, b);            // NB - 1 chunks ahead: the last slot is refilled while its successor is evaluated (below)
    const int kq = lane >> KSH;                              // the visible of the chunk this lane decides
    const int nchunks = (D + G - 1) / G;
    // what a chunk's passes read besides h: its eight w_dec rows -- visibles (2 p, 2 p + 1) side by side, one packed FMA serves both --, the lane's
    // b_dec and uniform.  (Measured and dropped: fetching them one chunk AHEAD, under the previous chunk's passes -- 47.0 vs 46.5 us per call: a lone
    // wave spends ~10 cycles per instruction of this chain whatever flies beside it.)
    struct ChunkIn { nade_f32x2 w[G / 2][4]; float bdv, u; };
    auto fetch_in = [&](int slot, int c, ChunkIn& ci) {
        const int i0 = c * G;
        // (Hn == 256, D % 8 != 0: the last chunk was copied from row D - 8 on -- visible i0 + k sits rsh rows further down; rows past D are never decided)
        const int rsh = (FULL && D >= G) ? i0 - min(i0, D - G) : 0;
        const float* rd = ring + (slot * 2 + 0) * G * 256 + rsh * 256;
#pragma unroll
        for (int p2 = 0; p2 < G / 2; ++p2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x0 = rd[(2 * p2) * 256 + off[q]], x1 = rd[(2 * p2 + 1) * 256 + off[q]];
                ci.w[p2][q] = nade_f32x2{in[q] ? x0 : 0.f, in[q] ? x1 : 0.f};
            }
        const int ivc = min(i0 + kq, D - 1);
        ci.bdv = sbd[ivc];
        ci.u = 0.f;
        if (TMODE != 0) {
            const uint32_t e_last = e0 + (uint32_t)min(i0 + G - 1, D - 1);
            if ((e_last >> 2) >= b0 + 64u) {                 // uniform: the window of 64 Philox blocks restarts at this chunk's first element
                b0 = (e0 + (uint32_t)i0) >> 2;
                refill();
            }
            ci.u = su[e0 + (uint32_t)ivc - 4u * b0];
        }
    };
    for (int c0 = 0; c0 < nchunks; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = c0 + b;
            if (c < nchunks) {                               // uniform
                // chunk c's 2 G copies are the oldest in flight; (NB - 2) chunks behind them may still be on their way
                SCH_TR(0, c);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
                SCH_TR(1, c);
                ChunkIn ci;
                fetch_in(b, c, ci);
                // the slot chunk c - 1 has just left takes chunk c + NB - 1: its 2 G copy instructions issue while the LDS reads above are on their way
                // (behind the passes, on their own, they were 200-280 ns of every chunk).  Always issued (rows clamped): the wait count stays a constant.
                stage((b + NB - 1) % NB, c + NB - 1);
                const int i0 = c * G;
                const int rsh = (FULL && D >= G) ? i0 - min(i0, D - G) : 0;
                const float* re = ring + (b * 2 + 1) * G * 256 + rsh * 256;
                const int iv = i0 + kq;
                const bool mine = iv < D;
                const float bdv = ci.bdv, u = ci.u;
                int sfrom = 0;                               // first undecided visible of the chunk
                SCH_TR(4, c);
                while (true) {
                    nade_f32x2 acc[G / 2];
#pragma unroll
                    for (int p2 = 0; p2 < G / 2; ++p2) {     // sum_j h_j w_dec[i0 + k][j], the contract's order (an IEEE fma per element, packed or not)
                        acc[p2] = nade_f32x2{0.f, 0.f};
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[p2] = __builtin_elementwise_fma(nade_f32x2{h[q], h[q]}, ci.w[p2][q], acc[p2]);
                    }
                    // reduce-scatter over the xor butterfly: at 32 the lower / upper half of the lanes keeps the lower / upper half of the visibles, at 16
                    // a quarter each, ... until a lane holds ONE visible (G = 8: after the xor-8 level, G = 16: after xor-4); the remaining levels add
                    // as in wave_xor_sum.  Every level adds the two numbers wave_xor_sum adds there.
                    nade_f32x2 v4[G / 4], v2[G / 8];
#pragma unroll
                    for (int j = 0; j < G / 4; ++j) {        // pair j (visibles 2 j, 2 j + 1) against pair j + G / 4 (G / 2 visibles further)
                        auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j].x), __float_as_uint(acc[j + G / 4].x), false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j].y), __float_as_uint(acc[j + G / 4].y), false, false);
                        v4[j] = nade_f32x2{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + nade_f32x2{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
                    }
#pragma unroll
                    for (int j = 0; j < G / 8; ++j) {
                        auto r0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v4[j].x), __float_as_uint(v4[j + G / 8].x), false, false);
                        auto r1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v4[j].y), __float_as_uint(v4[j + G / 8].y), false, false);
                        v2[j] = nade_f32x2{__uint_as_float(r0[0]), __uint_as_float(r1[0])} + nade_f32x2{__uint_as_float(r0[1]), __uint_as_float(r1[1])};
                    }
                    const bool up8 = (lane & 8) != 0;
                    float x;
                    if constexpr (G == 16) {                 // xor 8: pair 0 against pair 1; xor 4: the pair's two
                        const nade_f32x2 keep = up8 ? v2[1] : v2[0], give = up8 ? v2[0] : v2[1];
                        const nade_f32x2 v1 = keep + nade_f32x2{dpp_xor8(give.x), dpp_xor8(give.y)};
                        const bool up4 = (lane & 4) != 0;
                        const float k1 = up4 ? v1.y : v1.x, g1 = up4 ? v1.x : v1.y;
                        x = k1 + dpp_xor4(g1);
                    } else {
                        const float keep = up8 ? v2[0].y : v2[0].x, give = up8 ? v2[0].x : v2[0].y;
                        x = keep + dpp_xor8(give);
                        x = x + dpp_xor4(x);
                    }
                    x = x + dpp_xor2(x);
                    x = x + dpp_xor1(x);
                    const float l = bdv + x;
#ifdef SCH_TRACE
                    if (sfrom == 0) { asm volatile("" ::"v"(l)); SCH_TR(5, c); }
#endif
                    const bool open = mine && kq >= sfrom;   // lanes whose visible is still undecided
                    bool on;
                    if (TMODE != 0) {                        // u < det_sigmoid(l / T), settled by the hardware sigmoid unless within 1e-4 of a tie (draw_below)
                        const float xa = TMODE == 1 ? l : l / temperature;
                        const float r = sig_approx(xa);
                        const float d = u - r;
                        const bool tie = !(fabsf(d) > 1e-4f * r + 1e-30f);
                        if (__builtin_amdgcn_ballot_w64(tie && open) == 0ull) on = d < 0.f;
                        else on = u < det_sigmoid(xa);
                    } else {                                 // nade.py:278-279
                        const float d = sig_approx(l) - 0.5f;
                        const bool tie = !(fabsf(d) > 1e-4f);
                        if (__builtin_amdgcn_ballot_w64(tie && open) == 0ull) on = d > 0.f;
                        else on = det_sigmoid(l) >= 0.5f;
                    }
                    const uint64_t m = __builtin_amdgcn_ballot_w64(on && open) & (G == 16 ? 0x1111111111111111ull : 0x0101010101010101ull);
                    const int f = m != 0ull ? (int)(__builtin_ctzll(m) >> KSH) : G;     // the first draw of 1 (uniform)
#ifdef SCH_TRACE
                    if (sfrom == 0) { asm volatile("" ::"s"(f)); SCH_TR(6, c); }
#endif
                    if ((lane & ((1 << KSH) - 1)) == 0 && open && kq <= f) {
                        sp[iv] = l;
                        son[iv] = kq == f ? 1 : 0;
                    }
                    if (f >= G) break;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float x2 = re[f * 256 + off[q]];
                        a[q] = a[q] + (in[q] ? x2 : 0.f);
                        h[q] = det_sigmoid(a[q]);
                    }
                    sfrom = f + 1;
                    if (sfrom >= G || i0 + sfrom >= D) break;
                }
                SCH_TR(2, c);
                SCH_TR(3, c);
            }
        }
    }
    // after the scan: as nade_sample_kernel
    for (int i = lane; i < D; i += 64) {
        const float p = det_sigmoid(sp[i]);
        const bool on = son[i] != 0;
        samples[(size_t)row * s_row_stride + (size_t)i * s_elem_stride] = on ? 1 : 0;
        sp[i] = on ? logf(NADE_EPS + p) : logf(NADE_EPS + (1.0f - p));
    }
    if (nll != nullptr) {
        float logp = 0.f;
        for (int i = 0; i < D; ++i) logp += sp[i];
        if (lane == 0) nll[row] = -logp;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the last (clamped) copies land before the LDS is given back
}

static hipError_t sample_chunk_raise_lds() {                 // the ring + a row's scratch pass the 64 KB default of dynamic LDS
    static bool raised_[64];
    bool& raised = mnn_dev_flag(raised_);
    if (raised) return hipSuccess;
    hipError_t e = hipSuccess;
#define RAISE(TM, FU) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nade_sample_chunk_kernel<TM, FU, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); \
                      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nade_sample_chunk_kernel<TM, FU, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)
    RAISE(0, true); RAISE(1, true); RAISE(2, true); RAISE(0, false); RAISE(1, false); RAISE(2, false);
#undef RAISE
    raised = e == hipSuccess;
    return e;
}

static int launch_sample(hipStream_t st, const SampleJobs& J, int njobs, int N, int D, int Hn, float temperature, uint32_t row0, uint32_t sub,
                         long s_row_stride, int s_elem_stride) {
    const int tmode = temperature > 0.f ? (temperature == 1.0f ? 1 : 2) : 0;
    bool chunked = Hn % 4 == 0 && Hn >= 4 && getenv("MNN_SAMPLE_NO_CHUNK") == nullptr;      // (read per call: tests compare the two forms)
    for (int j = 0; j < njobs && chunked; ++j) chunked = (((uintptr_t)J.job[j].w_enc | (uintptr_t)J.job[j].w_dec) & 15) == 0;
    if (chunked) {                                           // eight visibles per pass, one wave per row
        // sixteen visibles per pass while every row has a CU of its own (96 KB of ring: one workgroup per CU); eight with more rows than that
        const int g = (long)N * njobs <= 256 && !getenv("MNN_SAMPLE_G8") ? 16 : 8;
        const size_t ldc = (size_t)sch_nb(g) * 2 * g * 1024 + (size_t)9 * ((D + 3) & ~3) + 1024;       // <= 96 KB + 13.5 KB + 1 KB (D <= 1536)
        MNN_HIP(sample_chunk_raise_lds());
#define SMC(TM, FU) do { if (g == 16) hipLaunchKernelGGL((nade_sample_chunk_kernel<TM, FU, 16>), dim3(N, njobs), dim3(64), ldc, st, J, N, D, Hn, temperature, row0, sub, s_row_stride, s_elem_stride); \
                         else hipLaunchKernelGGL((nade_sample_chunk_kernel<TM, FU, 8>), dim3(N, njobs), dim3(64), ldc, st, J, N, D, Hn, temperature, row0, sub, s_row_stride, s_elem_stride); } while (0)
        if (Hn == 256) { if (tmode == 0) SMC(0, true); else if (tmode == 1) SMC(1, true); else SMC(2, true); }
        else { if (tmode == 0) SMC(0, false); else if (tmode == 1) SMC(1, false); else SMC(2, false); }
#undef SMC
        MNN_LAUNCH_CHECK();
        return MNN_OK;
    }
    dim3 grid(cdiv(N, 4), njobs);
    const size_t lds = (size_t)36 * ((D + 3) & ~3) + 4096;   // 4 waves x ((2 f32 + u8) per visible + 256 uniforms)
#define SMP(TM, FU, SP) hipLaunchKernelGGL((nade_sample_kernel<TM, FU, SP>), grid, dim3(256), lds, st, J, N, D, Hn, temperature, row0, sub, s_row_stride, s_elem_stride)
    if (Hn == 256 && tmode == 1) { if (getenv("MNN_SAMPLE_NO_SPEC")) SMP(1, true, false); else SMP(1, true, true); }
    else if (Hn == 256) { if (tmode == 0) SMP(0, true, false); else SMP(2, true, false); }
    else { if (tmode == 0) SMP(0, false, false); else if (tmode == 1) SMP(1, false, false); else SMP(2, false, false); }
#undef SMP
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_nade_sample(mnn_stream_t s, int tracks, int N, int D, int Hn, const float* bias, int ld_bias, const float* w_enc,
                               const float* w_dec, float temperature, uint64_t seed, uint32_t row0, uint32_t sub, uint8_t* samples,
                               long s_track_stride, int s_row_stride, int s_elem_stride, float* nll) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_sample: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(bias && w_enc && w_dec && samples, "mnn_nade_sample: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_sample: ld_bias too small");
    MNN_REQUIRE(D <= 1536, "mnn_nade_sample: D <= 1536 (logits, b_dec and draws of a row are parked in LDS, 36 B per visible and wave; D=%d)", D);
    for (int m0 = 0; m0 < tracks; m0 += SAMPLE_MAX_JOBS) {            // the tracks of a MultiNADE: up to eight per launch
        SampleJobs J;
        memset(&J, 0, sizeof(J));
        const int nj = min(SAMPLE_MAX_JOBS, tracks - m0);
        for (int j = 0; j < nj; ++j) {
            const int m = m0 + j;
            J.job[j] = SampleJob{bias, ld_bias, m * Hn, tracks * Hn + m * D, w_enc + (size_t)m * D * Hn, w_dec + (size_t)m * D * Hn, seed,
                                 (uint32_t)(m * D), samples + (size_t)m * s_track_stride, nll ? nll + (size_t)m * N : nullptr};
        }
        const int rc = launch_sample((hipStream_t)s, J, nj, N, D, Hn, temperature, row0, sub, s_row_stride, s_elem_stride);
        if (rc != MNN_OK) return rc;
    }
    return MNN_OK;
}

// The same scan for SEVERAL single-NADE generators in one launch (the M per-track generators of a feedback-scan step,
// multinn_feedback.py:196: `generators[i].sample_single` for every track): job j has its own Dense output matrix, weights, seed and output
// pointer; rows, widths, temperature, the RNG row / sub counters and the output strides are shared.
extern "C" int mnn_nade_sample_multi(mnn_stream_t s, int njobs, const mnn_nade_sample_job* jobs, int N, int D, int Hn, float temperature,
                                     uint32_t row0, uint32_t sub, long s_row_stride, int s_elem_stride) {
    MNN_REQUIRE(njobs > 0 && njobs <= SAMPLE_MAX_JOBS && jobs && N > 0 && D > 0 && Hn > 0 && Hn <= 256 && D <= 1536,
                "mnn_nade_sample_multi: 1..%d jobs, N, D > 0, 0 < Hn <= 256, D <= 1536", SAMPLE_MAX_JOBS);
    SampleJobs J;
    memset(&J, 0, sizeof(J));
    for (int j = 0; j < njobs; ++j) {
        const mnn_nade_sample_job& q = jobs[j];
        MNN_REQUIRE(q.bias && q.w_enc && q.w_dec && q.samples && q.ld_bias >= Hn + D, "mnn_nade_sample_multi: job %d: null pointer or ld_bias < Hn + D", j);
        J.job[j] = SampleJob{q.bias, q.ld_bias, 0, Hn, q.w_enc, q.w_dec, q.seed, 0u, q.samples, q.nll};
    }
    return launch_sample((hipStream_t)s, J, njobs, N, D, Hn, temperature, row0, sub, s_row_stride, s_elem_stride);
}
