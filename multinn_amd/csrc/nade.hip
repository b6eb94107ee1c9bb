// NADE visible-order conditional scan for gfx950: log_prob forward, reverse-scan backward, sampling.
// Reference: /root/reference/multinn/models/common/nade.py:155-229 (log_prob), 231-308 (sample).
//
// Forward  (lane = row):     4 waves split the hidden units, each lane keeps its slice of the running
//                            pre-activation `a` in registers; w_enc/w_dec rows are wave-uniform scalar
//                            loads; the per-visible dot product needs NO cross-lane reduction, only one
//                            4-way cross-wave sum through LDS per block of 4 visibles.
// Backward (lane = hidden):  8 waves x 8 rows; the row is wave-uniform, so d l and v are scalars and the
//                            sums over rows for d w_dec / d w_enc accumulate in-lane; one LDS reduction
//                            + one 1 KiB-contiguous f32 atomic per visible per block.
// Sampling (one wave/row):   deterministic summation order + IEEE-only sigmoid so Bernoulli draws are
//                            bit-identical to oracle/det_ref.c.
#include "common.h"

#define NADE_EPS 1e-6f
#define LN2 0.6931471805599453f

__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * LN2; }

// ----------------------------------------------------------------------------------------------
// forward
// ----------------------------------------------------------------------------------------------
template <int HS>
__global__ void __launch_bounds__(256)
nade_fwd_kernel(int tracks, int N, int D, int Hn, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias,
                int ld_bias, const float* __restrict__ w_enc, const float* __restrict__ w_dec, const float* __restrict__ row_weight,
                float* __restrict__ nll, float* __restrict__ cond_p, float* __restrict__ d_bias) {
    __shared__ float red[2][4][4][64];
    __shared__ float red2[4][64];
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 64 + lane;
    const bool valid = row < N;
    const int rr = valid ? row : N - 1;
    const float* __restrict__ be = bias + (size_t)rr * ld_bias + m * Hn;
    const float* __restrict__ bd = bias + (size_t)rr * ld_bias + tracks * Hn + m * D;
    const uint8_t* __restrict__ vr = v + (size_t)m * v_track_stride + (size_t)rr * D;
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const float* __restrict__ wd = w_dec + (size_t)m * D * Hn;
    const int j0 = w * HS;

    // running pre-activation kept pre-scaled: a' = -log2(e) * a, so sigmoid(a) = rcp(1 + exp2(a'))
    float a[HS];
#pragma unroll
    for (int j = 0; j < HS; ++j) a[j] = (j0 + j < Hn) ? -MNN_LOG2E * be[j0 + j] : 0.f;

    const float rw = (row_weight != nullptr && valid) ? row_weight[row] : 0.f;
    float lp = 0.f;
    int buf = 0;
    for (int i0 = 0; i0 < D; i0 += 4) {
        float vi4[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) vi4[ib] = (i0 + ib < D) ? (float)vr[i0 + ib] : 0.f;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            const int i = i0 + ib;
            float acc = 0.f;
            if (i < D) {
                const float vs = -MNN_LOG2E * vi4[ib];
                const float* __restrict__ wdi = wd + (size_t)i * Hn + j0;
                const float* __restrict__ wei = we + (size_t)i * Hn + j0;
#pragma unroll
                for (int j = 0; j < HS; ++j) {
                    const bool in = j0 + j < Hn;
                    const float wdj = in ? wdi[j] : 0.f;
                    const float wej = in ? wei[j] : 0.f;
                    const float h = fast_rcp(1.0f + fast_exp2(a[j]));
                    acc = fmaf(h, wdj, acc);
                    a[j] = fmaf(vs, wej, a[j]);
                }
            }
            red[buf][w][ib][lane] = acc;
        }
        __syncthreads();
        const int i = i0 + w;          // wave w finalises visible i0 + w
        if (i < D) {
            const float l = bd[i] + ((red[buf][0][w][lane] + red[buf][1][w][lane]) + (red[buf][2][w][lane] + red[buf][3][w][lane]));
            const float p = fast_sigmoid(l);
            const float vi = w == 0 ? vi4[0] : (w == 1 ? vi4[1] : (w == 2 ? vi4[2] : vi4[3]));
            const float q = fast_sigmoid(-l);   // 1-p without cancellation (closer to the exact value than f32 `1 - p`)
            lp += vi > 0.5f ? fast_ln(NADE_EPS + p) : fast_ln(NADE_EPS + q);
            if (valid) {
                if (cond_p != nullptr) cond_p[((size_t)m * N + row) * D + i] = p;
                if (d_bias != nullptr) {
                    const float dnll_dp = vi > 0.5f ? -fast_rcp(NADE_EPS + p) : fast_rcp(NADE_EPS + q);
                    d_bias[(size_t)row * ld_bias + tracks * Hn + m * D + i] = rw * dnll_dp * p * q;
                }
            }
        }
        buf ^= 1;
    }
    red2[w][lane] = lp;
    __syncthreads();
    if (w == 0 && valid && nll != nullptr) nll[(size_t)m * N + row] = -((red2[0][lane] + red2[1][lane]) + (red2[2][lane] + red2[3][lane]));
}

extern "C" int mnn_nade_logprob_fwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                    const float* bias, int ld_bias, const float* w_enc, const float* w_dec, const float* row_weight,
                                    float* nll, float* cond_p, float* d_bias) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_logprob_fwd: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(v && bias && w_enc && w_dec, "mnn_nade_logprob_fwd: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_logprob_fwd: ld_bias %d < tracks*(Hn+D)", ld_bias);
    MNN_REQUIRE(d_bias == nullptr || row_weight != nullptr, "mnn_nade_logprob_fwd: d_bias needs row_weight");
    dim3 grid(cdiv(N, 64), tracks);
    hipStream_t st = (hipStream_t)s;
#define FWD(HS) hipLaunchKernelGGL(nade_fwd_kernel<HS>, grid, dim3(256), 0, st, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, \
                                   w_dec, row_weight, nll, cond_p, d_bias)
    if (Hn <= 64) FWD(16);
    else if (Hn <= 128) FWD(32);
    else FWD(64);
#undef FWD
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// backward: reverse scan over the visible order
// ----------------------------------------------------------------------------------------------
#define BWD_R 8   // rows per wave
template <int HQ>
__global__ void __launch_bounds__(512)
nade_bwd_kernel(int tracks, int N, int D, int Hn, const uint8_t* __restrict__ v, long v_track_stride, const float* __restrict__ bias,
                int ld_bias, const float* __restrict__ w_enc, const float* __restrict__ w_dec, float* __restrict__ d_bias,
                float* __restrict__ d_w_enc, float* __restrict__ d_w_dec) {
    __shared__ float red[2][8][2][HQ * 64];
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbase = blockIdx.x * 64 + w * BWD_R;
    const uint8_t* __restrict__ vm = v + (size_t)m * v_track_stride;
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const float* __restrict__ wd = w_dec + (size_t)m * D * Hn;
    const int dl_off = tracks * Hn + m * D;

    float a[BWD_R][HQ], G[BWD_R][HQ];
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = lane + 64 * q, row = rbase + r;
            a[r][q] = (row < N && j < Hn) ? bias[(size_t)row * ld_bias + m * Hn + j] : 0.f;
            G[r][q] = 0.f;
        }
    // a_D: replay the encoder updates in forward order (nade.py:219)
    for (int i = 0; i < D; ++i) {
        float wev[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) wev[q] = (lane + 64 * q < Hn) ? we[(size_t)i * Hn + lane + 64 * q] : 0.f;
#pragma unroll
        for (int r = 0; r < BWD_R; ++r) {
            const int row = rbase + r;
            if (row < N && vm[(size_t)row * D + i] != 0) {
#pragma unroll
                for (int q = 0; q < HQ; ++q) a[r][q] += wev[q];
            }
        }
    }
    int buf = 0;
    for (int i = D - 1; i >= 0; --i) {
        float wev[HQ], wdv[HQ], accd[HQ], acce[HQ];
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const bool in = lane + 64 * q < Hn;
            wev[q] = in ? we[(size_t)i * Hn + lane + 64 * q] : 0.f;
            wdv[q] = in ? wd[(size_t)i * Hn + lane + 64 * q] : 0.f;
            accd[q] = 0.f;
            acce[q] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < BWD_R; ++r) {
            const int row = rbase + r;
            if (row < N) {
                const float dl = d_bias[(size_t)row * ld_bias + dl_off + i];
                if (vm[(size_t)row * D + i] != 0) {
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        acce[q] += G[r][q];          // d w_enc[i] += v_i * G_{i+1}
                        a[r][q] -= wev[q];           // a_i = a_{i+1} - v_i * w_enc[i]
                    }
                }
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const float h = fast_sigmoid(a[r][q]);
                    accd[q] = fmaf(dl, h, accd[q]);
                    G[r][q] = fmaf(dl * wdv[q], fmaf(-h, h, h), G[r][q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            red[buf][w][0][lane + 64 * q] = accd[q];
            red[buf][w][1][lane + 64 * q] = acce[q];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * HQ * 64; e += 512) {
            const int which = e / (HQ * 64), j = e % (HQ * 64);
            if (j < Hn) {
                float sum = 0.f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) sum += red[buf][ww][which][j];
                atomicAdd((which == 0 ? d_w_dec : d_w_enc) + ((size_t)m * D + i) * Hn + j, sum);
            }
        }
        buf ^= 1;
    }
#pragma unroll
    for (int r = 0; r < BWD_R; ++r)
#pragma unroll
        for (int q = 0; q < HQ; ++q) {
            const int j = lane + 64 * q, row = rbase + r;
            if (row < N && j < Hn) d_bias[(size_t)row * ld_bias + m * Hn + j] = G[r][q];
        }
}

extern "C" int mnn_nade_logprob_bwd(mnn_stream_t s, int tracks, int N, int D, int Hn, const uint8_t* v, long v_track_stride,
                                    const float* bias, int ld_bias, const float* w_enc, const float* w_dec, float* d_bias, float* d_w_enc,
                                    float* d_w_dec) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_logprob_bwd: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(v && bias && w_enc && w_dec && d_bias && d_w_enc && d_w_dec, "mnn_nade_logprob_bwd: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_logprob_bwd: ld_bias too small");
    dim3 grid(cdiv(N, 64), tracks);
    hipStream_t st = (hipStream_t)s;
#define BWD(HQ) hipLaunchKernelGGL(nade_bwd_kernel<HQ>, grid, dim3(512), 0, st, tracks, N, D, Hn, v, v_track_stride, bias, ld_bias, w_enc, \
                                   w_dec, d_bias, d_w_enc, d_w_dec)
    if (Hn <= 64) BWD(1);
    else if (Hn <= 128) BWD(2);
    else BWD(4);
#undef BWD
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// sampling: one wave per (row, track); deterministic order (DESIGN.md "Deterministic sampling"):
//   lane l owns hidden j = l + 64 q, q = 0..3 ; acc_l = fma-chain over q ; xor-butterfly 32..1 ;
//   logit = b_dec + acc ; p = det_sigmoid(logit) ; draw = u < det_sigmoid(logit / T)
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
nade_sample_kernel(int tracks, int N, int D, int Hn, const float* __restrict__ bias, int ld_bias, const float* __restrict__ w_enc,
                   const float* __restrict__ w_dec, float temperature, uint64_t seed, uint32_t row0, uint32_t sub,
                   uint8_t* __restrict__ samples, long s_track_stride, int s_row_stride, int s_elem_stride, float* __restrict__ nll) {
    const int m = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;                                   // wave-uniform
    const float* __restrict__ we = w_enc + (size_t)m * D * Hn;
    const float* __restrict__ wd = w_dec + (size_t)m * D * Hn;
    const float* __restrict__ bd = bias + (size_t)row * ld_bias + tracks * Hn + m * D;
    float a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = (lane + 64 * q < Hn) ? bias[(size_t)row * ld_bias + m * Hn + lane + 64 * q] : 0.f;
    float logp = 0.f;
    for (int i = 0; i < D; ++i) {
        float acc = 0.f, wev[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = lane + 64 * q < Hn;
            const float wdq = in ? wd[(size_t)i * Hn + lane + 64 * q] : 0.f;
            wev[q] = in ? we[(size_t)i * Hn + lane + 64 * q] : 0.f;
            acc = fmaf(det_sigmoid(a[q]), wdq, acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc = acc + __shfl_xor(acc, o);
        const float l = bd[i] + acc;
        const float p = det_sigmoid(l);
        bool on;
        if (temperature > 0.f) {
            const float ps = temperature == 1.0f ? p : det_sigmoid(l / temperature);
            const float u = philox_uniform1(seed, MNN_STREAM_NADE, row0 + (uint32_t)row, sub, (uint32_t)(m * D + i));
            on = u < ps;
        } else {
            on = p >= 0.5f;                                  // nade.py:278-279
        }
        logp += on ? logf(NADE_EPS + p) : logf(NADE_EPS + (1.0f - p));
        if (on) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = a[q] + wev[q];
        }
        if (lane == 0) samples[(size_t)m * s_track_stride + (size_t)row * s_row_stride + (size_t)i * s_elem_stride] = on ? 1 : 0;
    }
    if (lane == 0 && nll != nullptr) nll[(size_t)m * N + row] = -logp;
}

extern "C" int mnn_nade_sample(mnn_stream_t s, int tracks, int N, int D, int Hn, const float* bias, int ld_bias, const float* w_enc,
                               const float* w_dec, float temperature, uint64_t seed, uint32_t row0, uint32_t sub, uint8_t* samples,
                               long s_track_stride, int s_row_stride, int s_elem_stride, float* nll) {
    MNN_REQUIRE(tracks > 0 && N > 0 && D > 0 && Hn > 0 && Hn <= 256, "mnn_nade_sample: need tracks,N,D>0 and 0<Hn<=256 (Hn=%d)", Hn);
    MNN_REQUIRE(bias && w_enc && w_dec && samples, "mnn_nade_sample: null pointer");
    MNN_REQUIRE(ld_bias >= tracks * (Hn + D), "mnn_nade_sample: ld_bias too small");
    dim3 grid(cdiv(N, 4), tracks);
    hipLaunchKernelGGL(nade_sample_kernel, grid, dim3(256), 0, (hipStream_t)s, tracks, N, D, Hn, bias, ld_bias, w_enc, w_dec, temperature,
                       seed, row0, sub, samples, s_track_stride, s_row_stride, s_elem_stride, nll);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
