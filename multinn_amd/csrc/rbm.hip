// RBM CD-k Gibbs chain, half-steps and free energy for gfx950.
// Reference: /root/reference/multinn/models/common/rbm.py:148-263, 337-387.
//
// One 256-thread block owns RBM_R rows for the whole chain: the visible and hidden states of those
// rows live in LDS (as f32 0/1), W is streamed from L2 (coalesced over the output unit; a transposed
// copy serves the visible half-step), each thread accumulates RBM_R rows of one output unit.
// Summation order is ascending input index with one fma per term, and the sigmoid uses IEEE ops
// only, so Bernoulli draws are bit-identical to oracle/det_ref.c.
#include "common.h"

#define RBM_R 8

__device__ __forceinline__ uint32_t rbm_rowid(const uint32_t* __restrict__ row_ids, uint32_t row0, int n) {
    return row_ids != nullptr ? row_ids[n] : row0 + (uint32_t)n;
}

// out-unit phase: for each output unit `o` (strided over threads) and each of the block's rows
//   z[r] = sum_{k asc} in[r][k] * Wk[k*ldw + o]  + bias[row r][o]
// in_s: LDS [RBM_R][Kpad] f32 (Kpad multiple of 4, zero padded).  Calls fn(r, o, z).
template <typename F>
__device__ __forceinline__ void rbm_phase(const float* __restrict__ in_s, int Kpad, int K, const float* __restrict__ Wk, int ldw, int n_out,
                                          F&& fn) {
    for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float acc[RBM_R];
#pragma unroll
        for (int r = 0; r < RBM_R; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 4) {
            float w[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = (k0 + kk < K) ? Wk[(size_t)(k0 + kk) * ldw + o] : 0.f;
#pragma unroll
            for (int r = 0; r < RBM_R; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(in_s + r * Kpad + k0);
                acc[r] = fmaf(x.x, w[0], acc[r]);
                acc[r] = fmaf(x.y, w[1], acc[r]);
                acc[r] = fmaf(x.z, w[2], acc[r]);
                acc[r] = fmaf(x.w, w[3], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RBM_R; ++r) fn(r, o, acc[r]);
    }
}

template <typename TV>
__device__ __forceinline__ void rbm_load_rows(const TV* __restrict__ src, int N, int n0, int K, int Kpad, float* __restrict__ dst_s) {
    for (int e = threadIdx.x; e < RBM_R * Kpad; e += blockDim.x) {
        const int r = e / Kpad, k = e % Kpad, n = n0 + r;
        dst_s[e] = (n < N && k < K) ? (float)src[(size_t)n * K + k] : 0.f;
    }
}

// ----------------------------------------------------------------------------------------------
// k-step Gibbs chain (rbm.py:192-231)
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rbm_gibbs_kernel(int N, int D, int Hn, int k, const uint8_t* __restrict__ v0, const float* __restrict__ W, const float* __restrict__ Wt,
                 const float* __restrict__ bh, int ld_bh, const float* __restrict__ bv, int ld_bv, uint64_t seed, uint32_t row0,
                 const uint32_t* __restrict__ row_ids, uint32_t sub0, float* __restrict__ p_v, uint8_t* __restrict__ v_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Dp = (D + 3) & ~3, Hp = (Hn + 3) & ~3;
    float* vs = smem;                 // [RBM_R][Dp]
    float* hs = smem + RBM_R * Dp;    // [RBM_R][Hp]
    const int n0 = blockIdx.x * RBM_R;
    rbm_load_rows<uint8_t>(v0, N, n0, D, Dp, vs);
    for (int e = threadIdx.x; e < RBM_R * Hp; e += blockDim.x) hs[e] = 0.f;
    __syncthreads();
    if (k == 0) {                     // tf.while_loop with zero iterations returns (v, v)
        for (int e = threadIdx.x; e < RBM_R * D; e += blockDim.x) {
            const int r = e / D, d = e % D, n = n0 + r;
            if (n < N) {
                if (p_v) p_v[(size_t)n * D + d] = vs[r * Dp + d];
                if (v_out) v_out[(size_t)n * D + d] = (uint8_t)vs[r * Dp + d];
            }
        }
        return;
    }
    for (int it = 0; it < k; ++it) {
        rbm_phase(vs, Dp, D, W, Hn, Hn, [&](int r, int j, float acc) {
            const int n = n0 + r;
            if (n >= N) return;
            const float p = det_sigmoid(acc + bh[(size_t)n * ld_bh + j]);
            const float u = philox_uniform1(seed, MNN_STREAM_RBM_H, rbm_rowid(row_ids, row0, n), sub0 + (uint32_t)it, (uint32_t)j);
            hs[r * Hp + j] = u < p ? 1.f : 0.f;
        });
        __syncthreads();
        const bool last = it == k - 1;
        rbm_phase(hs, Hp, Hn, Wt, D, D, [&](int r, int d, float acc) {
            const int n = n0 + r;
            if (n >= N) return;
            const float p = det_sigmoid(acc + bv[(size_t)n * ld_bv + d]);
            const float u = philox_uniform1(seed, MNN_STREAM_RBM_V, rbm_rowid(row_ids, row0, n), sub0 + (uint32_t)it, (uint32_t)d);
            const float s = u < p ? 1.f : 0.f;
            vs[r * Dp + d] = s;
            if (last) {
                if (p_v) p_v[(size_t)n * D + d] = p;
                if (v_out) v_out[(size_t)n * D + d] = (uint8_t)s;
            }
        });
        __syncthreads();
    }
}

extern "C" size_t mnn_rbm_workspace_bytes(int D, int Hn) { return (size_t)D * Hn * sizeof(float); }

extern "C" int mnn_transpose(mnn_stream_t s, const void* in, int in_dtype, int R, int C, int ld_in, void* out, int out_dtype, int ld_out);

static size_t rbm_lds_bytes(int D, int Hn) { return (size_t)RBM_R * (((D + 3) & ~3) + ((Hn + 3) & ~3)) * sizeof(float); }

extern "C" int mnn_rbm_gibbs(mnn_stream_t s, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh,
                             const float* bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0, float* p_v,
                             uint8_t* v_out, void* workspace) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && k >= 0, "mnn_rbm_gibbs: bad sizes N=%d D=%d Hn=%d k=%d", N, D, Hn, k);
    MNN_REQUIRE(v0 && W && bh && bv && workspace, "mnn_rbm_gibbs: null pointer");
    MNN_REQUIRE((ld_bh == 0 || ld_bh >= Hn) && (ld_bv == 0 || ld_bv >= D), "mnn_rbm_gibbs: bad bias leading dimension");
    MNN_REQUIRE(rbm_lds_bytes(D, Hn) <= 160 * 1024, "mnn_rbm_gibbs: D+Hn too large for LDS");
    int rc = mnn_transpose(s, W, MNN_F32, D, Hn, Hn, workspace, MNN_F32, D);
    if (rc != MNN_OK) return rc;
    hipLaunchKernelGGL(rbm_gibbs_kernel, dim3(cdiv(N, RBM_R)), dim3(256), rbm_lds_bytes(D, Hn), (hipStream_t)s, N, D, Hn, k, v0, W,
                       (const float*)workspace, bh, ld_bh, bv, ld_bv, seed, row0, row_ids, sub0, p_v, v_out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// half-steps (rbm.py:148-190) -- also the DBN encode / decode steps (dbn.py:136-180)
// ----------------------------------------------------------------------------------------------
template <typename TV>
__global__ void __launch_bounds__(256)
rbm_half_kernel(int N, int K, int n_out, const TV* __restrict__ in, const float* __restrict__ Wk, int ldw, const float* __restrict__ b, int ld_b,
                int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* __restrict__ p_out, uint8_t* __restrict__ s_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Kp = (K + 3) & ~3;
    const int n0 = blockIdx.x * RBM_R;
    rbm_load_rows<TV>(in, N, n0, K, Kp, smem);
    __syncthreads();
    rbm_phase(smem, Kp, K, Wk, ldw, n_out, [&](int r, int o, float acc) {
        const int n = n0 + r;
        if (n >= N) return;
        const float p = det_sigmoid(acc + b[(size_t)n * ld_b + o]);
        if (p_out) p_out[(size_t)n * n_out + o] = p;
        if (s_out) {
            const float u = philox_uniform1(seed, (uint32_t)stream_id, row0 + (uint32_t)n, sub, (uint32_t)o);
            s_out[(size_t)n * n_out + o] = u < p ? 1 : 0;
        }
    });
}

static int launch_half(hipStream_t st, int N, int K, int n_out, const void* in, int in_dtype, const float* Wk, int ldw, const float* b,
                       int ld_b, int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_out, uint8_t* s_out) {
    const size_t lds = (size_t)RBM_R * ((K + 3) & ~3) * sizeof(float);
    if (in_dtype == MNN_U8)
        hipLaunchKernelGGL(rbm_half_kernel<uint8_t>, dim3(cdiv(N, RBM_R)), dim3(256), lds, st, N, K, n_out, (const uint8_t*)in, Wk, ldw, b, ld_b,
                           stream_id, seed, row0, sub, p_out, s_out);
    else
        hipLaunchKernelGGL(rbm_half_kernel<float>, dim3(cdiv(N, RBM_R)), dim3(256), lds, st, N, K, n_out, (const float*)in, Wk, ldw, b, ld_b,
                           stream_id, seed, row0, sub, p_out, s_out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_rbm_hidden(mnn_stream_t s, int N, int D, int Hn, const void* v, int v_dtype, const float* W, const float* bh, int ld_bh,
                              int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_h, uint8_t* h) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && v && W && bh && (p_h || h), "mnn_rbm_hidden: bad arguments");
    MNN_REQUIRE(v_dtype == MNN_U8 || v_dtype == MNN_F32, "mnn_rbm_hidden: v dtype must be u8/f32");
    MNN_REQUIRE(ld_bh == 0 || ld_bh >= Hn, "mnn_rbm_hidden: bad ld_bh");
    return launch_half((hipStream_t)s, N, D, Hn, v, v_dtype, W, Hn, bh, ld_bh, stream_id, seed, row0, sub, p_h, h);
}

extern "C" int mnn_rbm_visible(mnn_stream_t s, int N, int D, int Hn, const void* h, int h_dtype, const float* W, const float* bv, int ld_bv,
                               int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_v, uint8_t* v, void* workspace) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && h && W && bv && workspace && (p_v || v), "mnn_rbm_visible: bad arguments");
    MNN_REQUIRE(h_dtype == MNN_U8 || h_dtype == MNN_F32, "mnn_rbm_visible: h dtype must be u8/f32");
    MNN_REQUIRE(ld_bv == 0 || ld_bv >= D, "mnn_rbm_visible: bad ld_bv");
    int rc = mnn_transpose(s, W, MNN_F32, D, Hn, Hn, workspace, MNN_F32, D);
    if (rc != MNN_OK) return rc;
    return launch_half((hipStream_t)s, N, Hn, D, h, h_dtype, (const float*)workspace, D, bv, ld_bv, stream_id, seed, row0, sub, p_v, v);
}

// ----------------------------------------------------------------------------------------------
// free energy, per row (rbm.py:256-258; R4):  F[n] = -sum_j softplus((vW)_j + bh[n,j]) - v.bv[n]
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }

__global__ void __launch_bounds__(256)
rbm_free_energy_kernel(int N, int D, int Hn, const uint8_t* __restrict__ v, const float* __restrict__ W, const float* __restrict__ bh, int ld_bh,
                       const float* __restrict__ bv, int ld_bv, float* __restrict__ F) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float fsum[RBM_R];
    const int Dp = (D + 3) & ~3;
    const int n0 = blockIdx.x * RBM_R;
    rbm_load_rows<uint8_t>(v, N, n0, D, Dp, smem);
    if (threadIdx.x < RBM_R) fsum[threadIdx.x] = 0.f;
    __syncthreads();
    float part[RBM_R];
#pragma unroll
    for (int r = 0; r < RBM_R; ++r) part[r] = 0.f;
    rbm_phase(smem, Dp, D, W, Hn, Hn, [&](int r, int j, float acc) {
        const int n = n0 + r;
        if (n < N) part[r] -= softplus_f(acc + bh[(size_t)n * ld_bh + j]);
    });
    for (int d = threadIdx.x; d < D; d += blockDim.x)
#pragma unroll
        for (int r = 0; r < RBM_R; ++r) {
            const int n = n0 + r;
            if (n < N) part[r] -= smem[r * Dp + d] * bv[(size_t)n * ld_bv + d];
        }
#pragma unroll
    for (int r = 0; r < RBM_R; ++r) {
        float x = part[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&fsum[r], x);
    }
    __syncthreads();
    if (threadIdx.x < RBM_R && n0 + threadIdx.x < N) F[n0 + threadIdx.x] = fsum[threadIdx.x];
}

extern "C" int mnn_rbm_free_energy(mnn_stream_t s, int N, int D, int Hn, const uint8_t* v, const float* W, const float* bh, int ld_bh,
                                   const float* bv, int ld_bv, float* F) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && v && W && bh && bv && F, "mnn_rbm_free_energy: bad arguments");
    MNN_REQUIRE((ld_bh == 0 || ld_bh >= Hn) && (ld_bv == 0 || ld_bv >= D), "mnn_rbm_free_energy: bad bias leading dimension");
    const size_t lds = (size_t)RBM_R * ((D + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(rbm_free_energy_kernel, dim3(cdiv(N, RBM_R)), dim3(256), lds, (hipStream_t)s, N, D, Hn, v, W, bh, ld_bh, bv, ld_bv, F);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
