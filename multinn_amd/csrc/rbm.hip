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
// The bias rows bias[(n0 + r) * ld_bias + o] (ld_bias = 0: one shared row) are REQUESTED in front of the K loop and added behind it; the
// weights come 16 k at a time, unconditionally (k clamped: the inputs are zero past K, so the clamped weight contributes fma(0, w, acc) = acc
// exactly -- same ascending fma chain, bit for bit).  Round 3: with `k < K ? load : 0` per weight and the bias loaded inside the per-row
// callback every load was waited for on its own (s_waitcnt vmcnt(0) per element: 4 + 8 memory round trips per 4 k).
template <typename F>
__device__ __forceinline__ void rbm_phase(const float* __restrict__ in_s, int Kpad, int K, const float* __restrict__ Wk, int ldw, int n_out,
                                          const float* __restrict__ bias, int ld_bias, int n0, int N, F&& fn) {
    for (int o = threadIdx.x; o < n_out; o += blockDim.x) {
        float acc[RBM_R], bb[RBM_R];
#pragma unroll
        for (int r = 0; r < RBM_R; ++r) { acc[r] = 0.f; bb[r] = bias[(size_t)min(n0 + r, N - 1) * ld_bias + o]; }
        for (int k0 = 0; k0 < K; k0 += 16) {
            float w[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) w[kk] = Wk[(size_t)min(k0 + kk, K - 1) * ldw + o];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (k0 + 4 * q < K) {                       // block-uniform; Kpad covers the quad
#pragma unroll
                    for (int r = 0; r < RBM_R; ++r) {
                        const float4 x = *reinterpret_cast<const float4*>(in_s + r * Kpad + k0 + 4 * q);
                        acc[r] = fmaf(x.x, w[4 * q + 0], acc[r]);
                        acc[r] = fmaf(x.y, w[4 * q + 1], acc[r]);
                        acc[r] = fmaf(x.z, w[4 * q + 2], acc[r]);
                        acc[r] = fmaf(x.w, w[4 * q + 3], acc[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RBM_R; ++r) fn(r, o, acc[r] + bb[r]);
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
                 const uint32_t* __restrict__ row_ids, uint32_t sub0, float* __restrict__ p_v, uint8_t* __restrict__ v_out,
                 const int* __restrict__ seed_step) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (seed_step != nullptr) seed += (uint64_t)(int64_t)*seed_step;      // step counter on the device: a captured launch draws anew every replay
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
        rbm_phase(vs, Dp, D, W, Hn, Hn, bh, ld_bh, n0, N, [&](int r, int j, float z) {
            const int n = n0 + r;
            if (n >= N) return;
            const float p = det_sigmoid(z);
            const float u = philox_uniform1(seed, MNN_STREAM_RBM_H, rbm_rowid(row_ids, row0, n), sub0 + (uint32_t)it, (uint32_t)j);
            hs[r * Hp + j] = u < p ? 1.f : 0.f;
        });
        __syncthreads();
        const bool last = it == k - 1;
        rbm_phase(hs, Hp, Hn, Wt, D, D, bv, ld_bv, n0, N, [&](int r, int d, float z) {
            const int n = n0 + r;
            if (n >= N) return;
            const float p = det_sigmoid(z);
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

// ----------------------------------------------------------------------------------------------
// The same chain with W RESIDENT IN LDS (D (Hn + 1) floats fit: D = 88, Hn = 256 is 90 KB).  The streaming kernel above fetches every
// W row from L2 inside the k loop, twice per Gibbs iteration, and waits for it: 33 us per iteration whatever the row count.  Here W is
// read once per workgroup; the row stride Hn + 1 makes both walks conflict-free (hidden phase: consecutive threads, consecutive
// columns; visible phase: thread d walks row d, bank (d + k) mod 32), so no transposed copy either.  R = 2 rows per workgroup (many
// short workgroups; used below 2048 rows, see mnn_rbm_gibbs); a phase with fewer outputs than threads splits the rows over the
// spare threads (visible phase at D = 88: two row groups).  Biases stay in registers over the chain.  Arithmetic and order are
// the streaming kernel's: ascending-index fma chain from 0, + bias, det_sigmoid, Philox draw -- bit-identical draws.
// ----------------------------------------------------------------------------------------------
template <int R, int RG, typename F>      // RG rows per thread; thread t -> (row group t / n_out, output t % n_out)
__device__ __forceinline__ void rbm_phase_lds(const float* __restrict__ in_s, int Kpad, int K, const float* __restrict__ Ws, int w_k_stride,
                                              int w_o_stride, int n_out, F&& fn) {
    const int g = threadIdx.x / n_out, o = threadIdx.x - g * n_out;
    if (g >= R / RG) return;
    const float* __restrict__ wp = Ws + (size_t)o * w_o_stride;
    const float* __restrict__ xp = in_s + (size_t)g * RG * Kpad;
    float acc[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 4) {
        float w[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) w[kk] = wp[(size_t)min(k0 + kk, K - 1) * w_k_stride];     // beyond K the input is the zero padding
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(xp + r * Kpad + k0);
            acc[r] = fmaf(x.x, w[0], acc[r]);
            acc[r] = fmaf(x.y, w[1], acc[r]);
            acc[r] = fmaf(x.z, w[2], acc[r]);
            acc[r] = fmaf(x.w, w[3], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < RG; ++r) fn(g * RG + r, r, o, acc[r]);
}

template <int R, int RGH, int RGV>        // rows per thread in the hidden / visible phase (R / RG row groups of n_out threads each)
__global__ void __launch_bounds__(256)
rbm_gibbs_lds_kernel(int N, int D, int Hn, int k, const uint8_t* __restrict__ v0, const float* __restrict__ W, const float* __restrict__ bh,
                     int ld_bh, const float* __restrict__ bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* __restrict__ row_ids,
                     uint32_t sub0, float* __restrict__ p_v, uint8_t* __restrict__ v_out, const int* __restrict__ seed_step) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (seed_step != nullptr) seed += (uint64_t)(int64_t)*seed_step;
    const int Dp = (D + 3) & ~3, Hp = (Hn + 3) & ~3, ldw = Hn + 1;
    float* vs = smem;                 // [R][Dp]
    float* hs = vs + R * Dp;          // [R][Hp]
    float* Ws = hs + R * Hp;          // [D][ldw]
    const int n0 = blockIdx.x * R;
    for (int d = threadIdx.x >> 6; d < D; d += 4)            // one wave per row of W: coalesced, no index division
        for (int j = threadIdx.x & 63; j < Hn; j += 64) Ws[d * ldw + j] = W[(size_t)d * Hn + j];
    for (int e = threadIdx.x; e < R * Dp; e += blockDim.x) {
        const int r = e / Dp, kx = e % Dp, n = n0 + r;
        vs[e] = (n < N && kx < D) ? (float)v0[(size_t)n * D + kx] : 0.f;
    }
    for (int e = threadIdx.x; e < R * Hp; e += blockDim.x) hs[e] = 0.f;
    // this thread's biases and row ids: constant over the chain
    float bhr[RGH], bvr[RGV];
    uint32_t idh[RGH], idv[RGV];
    {
        const int g = threadIdx.x / Hn, o = threadIdx.x - g * Hn;
#pragma unroll
        for (int r = 0; r < RGH; ++r) {
            const int n = min(n0 + g * RGH + r, N - 1);
            bhr[r] = bh[(size_t)n * ld_bh + min(o, Hn - 1)];
            idh[r] = rbm_rowid(row_ids, row0, n);
        }
    }
    {
        const int g = threadIdx.x / D, o = threadIdx.x - g * D;
#pragma unroll
        for (int r = 0; r < RGV; ++r) {
            const int n = min(n0 + min(g * RGV + r, R - 1), N - 1);
            bvr[r] = bv[(size_t)n * ld_bv + min(o, D - 1)];
            idv[r] = rbm_rowid(row_ids, row0, n);
        }
    }
    __syncthreads();
    if (k == 0) {                     // tf.while_loop with zero iterations returns (v, v)
        for (int e = threadIdx.x; e < R * D; e += blockDim.x) {
            const int r = e / D, d = e % D, n = n0 + r;
            if (n < N) {
                if (p_v) p_v[(size_t)n * D + d] = vs[r * Dp + d];
                if (v_out) v_out[(size_t)n * D + d] = (uint8_t)vs[r * Dp + d];
            }
        }
        return;
    }
    for (int it = 0; it < k; ++it) {
        rbm_phase_lds<R, RGH>(vs, Dp, D, Ws, ldw, 1, Hn, [&](int r, int rl, int j, float acc) {
            if (n0 + r >= N) return;
            const float p = det_sigmoid(acc + bhr[rl]);
            const float u = philox_uniform1(seed, MNN_STREAM_RBM_H, idh[rl], sub0 + (uint32_t)it, (uint32_t)j);
            hs[r * Hp + j] = u < p ? 1.f : 0.f;
        });
        __syncthreads();
        const bool last = it == k - 1;
        rbm_phase_lds<R, RGV>(hs, Hp, Hn, Ws, 1, ldw, D, [&](int r, int rl, int d, float acc) {
            const int n = n0 + r;
            if (n >= N) return;
            const float p = det_sigmoid(acc + bvr[rl]);
            const float u = philox_uniform1(seed, MNN_STREAM_RBM_V, idv[rl], sub0 + (uint32_t)it, (uint32_t)d);
            const float sv = u < p ? 1.f : 0.f;
            vs[r * Dp + d] = sv;
            if (last) {
                if (p_v) p_v[(size_t)n * D + d] = p;
                if (v_out) v_out[(size_t)n * D + d] = (uint8_t)sv;
            }
        });
        __syncthreads();
    }
}

static size_t rbm_lds_resident_bytes(int R, int D, int Hn) {
    return ((size_t)R * (((D + 3) & ~3) + ((Hn + 3) & ~3)) + (size_t)D * (Hn + 1)) * sizeof(float);
}

// Launch the resident-W form when it applies (both phases fit 256 threads, W fits LDS); false: the caller streams.
template <int R, int RGH, int RGV>
static bool launch_gibbs_lds(hipStream_t st, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh, const float* bv,
                             int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0, float* p_v, uint8_t* v_out,
                             const int* seed_step) {
    const size_t lds = rbm_lds_resident_bytes(R, D, Hn);
    static bool raised_[64];                           // per instantiation and device: dynamic LDS above 64 KB has to be asked for once
    bool& raised = mnn_dev_flag(raised_);
    if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&rbm_gibbs_lds_kernel<R, RGH, RGV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        raised = true;
    }
    hipLaunchKernelGGL((rbm_gibbs_lds_kernel<R, RGH, RGV>), dim3(cdiv(N, R)), dim3(256), lds, st, N, D, Hn, k, v0, W, bh, ld_bh, bv, ld_bv, seed,
                       row0, row_ids, sub0, p_v, v_out, seed_step);
    return true;
}

// ----------------------------------------------------------------------------------------------
// The chain on the MATRIX CORES, bit for bit the same draws (training batches, N >= 2048 rows).  v_mfma_f32_32x32x2_f32 computes
// D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) with one IEEE rounding per product-add (cdna_hip_programming.md, "FP32-input MFMA"): a run of
// such instructions over ascending k IS the ascending-index fmaf chain of the vector kernels above, so the logits -- and with them every
// Bernoulli draw -- are identical, at the matrix pipe's rate instead of one fma per lane and term (the vector form reaches 22 TFLOP/s of the
// 157 f32 peak: its inner loop is LDS reads and address arithmetic).  Layout: a workgroup (8 waves) owns 64 rows for the whole chain; W sits
// in LDS once (f32 [D][Hn + 1]); the binary v / h states sit in LDS as BYTES (row pitch = an odd number of words: the B-operand reads of the
// 32 rows of a tile hit 32 banks).  The product is formed TRANSPOSED, C[out unit][row] = sum_k W(k, unit) state[row][k] (A = the weights,
// B = the states): a lane then holds four CONSECUTIVE output units of one row per accumulator quad = exactly the four uniforms of one
// Philox block (element >> 2 is the block counter), so every Philox evaluation is used in full -- the vector kernels draw one element per
// evaluation.  Hidden phase: 2 row tiles x (Hn / 32) unit tiles, two unit tiles per wave share the state operand; visible phase:
// 2 x ceil(D / 32) jobs on the first waves (one K = Hn chain per output: it cannot be split without changing the summation order).
// ----------------------------------------------------------------------------------------------
typedef float gm_f32x16 __attribute__((ext_vector_type(16)));
#define GM_ROWS 64

struct GibbsMfmaArgs {
    int N, D, Hn, k;
    const uint8_t* v0; const float* W; const float* bh; int ld_bh; const float* bv; int ld_bv;
    uint64_t seed; uint32_t row0; const uint32_t* row_ids; uint32_t sub0; float* p_v; uint8_t* v_out; const int* seed_step;
};

// pitch (bytes) of a byte-state row of `n` cells: covers n rounded up to even (the k pairs of the MFMA), a whole number of words, and an ODD
// number of words (rows land in distinct banks)
static __host__ __device__ __forceinline__ int gm_pitch(int n) { int w = (n + 1 + 3) / 4; return 4 * (w | 1); }

// one output tile (32 units x 32 rows) of a phase: K ascending in pairs; A = W (unit, k) from LDS, B = the rows' byte states
// (Reading the operands of eight k-pairs ahead of their MFMAs -- what the single-wave det-step kernels need -- was measured SLOWER here, 2.78 ->
// 3.19 ms per jamming step: with two waves per SIMD the other wave's MFMA fills the LDS wait, and the batches cost 60 more registers.)
template <bool VIS>
__device__ __forceinline__ void gm_chain(const float* __restrict__ Ws, int ldw, const uint8_t* __restrict__ st, int pitch, int K, int unit,
                                         int lane, gm_f32x16& acc) {
    const int r = lane & 31, hh = lane >> 5;
    const uint8_t* sp = st + r * pitch + hh;
    // hidden phase: A[i = hidden j][k = d] = W[d][j] (walks down a column); visible phase: A[i = visible d][k = j] = W[d][j] (walks a row)
    const float* ap = VIS ? Ws + (size_t)unit * ldw + hh : Ws + (size_t)hh * ldw + unit;
    const int astep = VIS ? 2 : 2 * ldw;
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(size_t)s * astep], (float)sp[2 * s], acc, 0, 0, 0);
}

__global__ void __launch_bounds__(512) rbm_gibbs_mfma_kernel(GibbsMfmaArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    uint64_t seed = A.seed;
    if (A.seed_step != nullptr) seed += (uint64_t)(int64_t)*A.seed_step;
    const int N = A.N, D = A.D, Hn = A.Hn, ldw = Hn + 1;
    const int De = (D + 1) & ~1, He = (Hn + 1) & ~1;          // K of the two phases (even; the states are zero past D / Hn)
    const int pv = gm_pitch(D), ph = gm_pitch(Hn);
    float* Ws = smem;                                         // [De][ldw] (row D, if any, repeats row D - 1: its inputs are zero)
    uint8_t* vs = reinterpret_cast<uint8_t*>(Ws + (size_t)De * ldw);      // [64][pv]
    uint8_t* hs = vs + GM_ROWS * pv;                                        // [64][ph]
    const int n0 = blockIdx.x * GM_ROWS;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int d = w; d < De; d += 8)                          // one wave per row of W: coalesced
        for (int j = lane; j < ldw; j += 64) Ws[d * ldw + j] = j < Hn ? A.W[(size_t)min(d, D - 1) * Hn + j] : 0.f;
    {   // v0 rows: thread t -> row t >> 3, eight lanes walk its bytes, sixteen loads in flight (unconditional, clamped)
        const int rr = threadIdx.x >> 3, sub = threadIdx.x & 7, n = n0 + rr;
        const uint8_t* __restrict__ src = A.v0 + (size_t)min(n, N - 1) * D;
        for (int kb = sub; kb < pv; kb += 128) {
            uint8_t v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = src[min(kb + 8 * q, D - 1)];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (kb + 8 * q < pv) vs[rr * pv + kb + 8 * q] = (n < N && kb + 8 * q < D) ? v[q] : (uint8_t)0;
        }
    }
    for (int e = threadIdx.x; e < GM_ROWS * ph; e += 512) hs[e] = 0;
    const int r = lane & 31, hh = lane >> 5;
    // jobs: hidden -- row tile w >> 2, unit tiles 2 (w & 3) + 8 q ... (two per pass, all Hn / 32 covered in ceil(Hn / 256) passes);
    //       visible -- job id w (+ 8 per pass) = row tile * ndt + unit tile
    const int nht = (Hn + 31) / 32, ndt = (D + 31) / 32;
    const int rt_h = w >> 2;
    const int row_h = n0 + 32 * rt_h + r;                     // the batch row of this lane's accumulator column (hidden jobs)
    const uint32_t id_h = rbm_rowid(A.row_ids, A.row0, min(row_h, N - 1));
    // the biases of this wave's first-pass jobs stay in registers over the chain (they do not change between Gibbs iterations; loaded inside
    // the epilogue, every accumulator quad waited for its own L2 round trip in every iteration)
    float bhr[2][16], bvr[16];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int j = min(32 * (2 * (w & 3) + q) + (e & 3) + 8 * (e >> 2) + 4 * hh, Hn - 1);
            bhr[q][e] = A.bh[(size_t)min(row_h, N - 1) * A.ld_bh + j];
        }
    {
        const int job = min(w, 2 * ndt - 1), rt = job / ndt, dt = job - rt * ndt;
        const int row = min(n0 + 32 * rt + r, N - 1);
#pragma unroll
        for (int e = 0; e < 16; ++e) bvr[e] = A.bv[(size_t)row * A.ld_bv + min(32 * dt + (e & 3) + 8 * (e >> 2) + 4 * hh, D - 1)];
    }
    __syncthreads();
    if (A.k == 0) {
        for (int e = threadIdx.x; e < GM_ROWS * D; e += 512) {
            const int rr = e / D, d = e - rr * D, n = n0 + rr;
            if (n < N) {
                if (A.p_v) A.p_v[(size_t)n * D + d] = (float)vs[rr * pv + d];
                if (A.v_out) A.v_out[(size_t)n * D + d] = vs[rr * pv + d];
            }
        }
        return;
    }
    for (int it = 0; it < A.k; ++it) {
        // ---- hidden phase ----
        for (int jt0 = 2 * (w & 3); jt0 < nht; jt0 += 8) {
            gm_f32x16 acc[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
            const uint8_t* sp = vs + (32 * rt_h + r) * pv + hh;
            const int u0 = min(32 * jt0 + r, Hn - 1), u1 = min(32 * (jt0 + 1) + r, Hn - 1);
            const float* a0 = Ws + (size_t)hh * ldw + u0;
            const float* a1 = Ws + (size_t)hh * ldw + u1;
            const bool two = jt0 + 1 < nht;
            for (int s = 0; s < De / 2; ++s) {                // the two unit tiles share the state operand
                const float b = (float)sp[2 * s];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[(size_t)s * 2 * ldw], b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[(size_t)s * 2 * ldw], b, acc[1], 0, 0, 0);      // (a lone last tile repeats column Hn - 1: discarded)
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && !two) break;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int j0 = 32 * (jt0 + q) + 8 * g4 + 4 * hh;          // four consecutive hidden units: one Philox block
                    if (j0 >= Hn) continue;
                    float u[4];
                    philox_uniform4(seed, MNN_STREAM_RBM_H, id_h, A.sub0 + (uint32_t)it, (uint32_t)(j0 >> 2), u);
                    uint32_t pk = 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = min(j0 + e, Hn - 1);
                        const float bb = jt0 < 8 ? bhr[q][4 * g4 + e] : A.bh[(size_t)min(row_h, N - 1) * A.ld_bh + j];
                        const float p = det_sigmoid(acc[q][4 * g4 + e] + bb);
                        pk |= (u[e] < p && j0 + e < Hn ? 1u : 0u) << (8 * e);
                    }
                    *reinterpret_cast<uint32_t*>(hs + (32 * rt_h + r) * ph + j0) = pk;     // j0 % 4 == 0, ph % 4 == 0
                }
            }
        }
        __syncthreads();
        // ---- visible phase ----
        const bool last = it == A.k - 1;
        for (int job = w; job < 2 * ndt; job += 8) {
            const int rt = job / ndt, dt = job - rt * ndt;
            gm_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            gm_chain<true>(Ws, ldw, hs + 32 * rt * ph, ph, He, min(32 * dt + r, D - 1), lane, acc);
            const int row = n0 + 32 * rt + r;
            const uint32_t idv = rbm_rowid(A.row_ids, A.row0, min(row, N - 1));
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = 32 * dt + 8 * g4 + 4 * hh;
                if (d0 >= D) continue;
                float u[4];
                philox_uniform4(seed, MNN_STREAM_RBM_V, idv, A.sub0 + (uint32_t)it, (uint32_t)(d0 >> 2), u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = d0 + e;
                    if (d >= D) continue;
                    const float p = det_sigmoid(acc[4 * g4 + e] + (job < 8 ? bvr[4 * g4 + e] : A.bv[(size_t)min(row, N - 1) * A.ld_bv + d]));
                    const uint8_t sv = u[e] < p ? 1 : 0;
                    vs[(32 * rt + r) * pv + d] = sv;
                    if (last && row < N) {
                        if (A.p_v) A.p_v[(size_t)row * D + d] = p;
                        if (A.v_out) A.v_out[(size_t)row * D + d] = sv;
                    }
                }
            }
        }
        __syncthreads();
    }
}

static size_t gibbs_mfma_lds_bytes(int D, int Hn) {
    return (size_t)((D + 1) & ~1) * (Hn + 1) * sizeof(float) + (size_t)GM_ROWS * (gm_pitch(D) + gm_pitch(Hn));
}

extern "C" size_t mnn_rbm_workspace_bytes(int D, int Hn) { return (size_t)D * Hn * sizeof(float); }

extern "C" int mnn_transpose(mnn_stream_t s, const void* in, int in_dtype, int R, int C, int ld_in, void* out, int out_dtype, int ld_out);

static size_t rbm_lds_bytes(int D, int Hn) { return (size_t)RBM_R * (((D + 3) & ~3) + ((Hn + 3) & ~3)) * sizeof(float); }

extern "C" int mnn_rbm_gibbs_stepped(mnn_stream_t s, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh,
                                     const float* bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0, float* p_v,
                                     uint8_t* v_out, void* workspace, const int* seed_step) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && k >= 0, "mnn_rbm_gibbs: bad sizes N=%d D=%d Hn=%d k=%d", N, D, Hn, k);
    MNN_REQUIRE(v0 && W && bh && bv && workspace, "mnn_rbm_gibbs: null pointer");
    MNN_REQUIRE((ld_bh == 0 || ld_bh >= Hn) && (ld_bv == 0 || ld_bv >= D), "mnn_rbm_gibbs: bad bias leading dimension");
    MNN_REQUIRE(rbm_lds_bytes(D, Hn) <= 160 * 1024, "mnn_rbm_gibbs: D+Hn too large for LDS");
    if (N < 2048 && Hn <= 256 && D <= 256 && getenv("MNN_RBM_STREAM_W") == nullptr) {
        // sampling-sized batches: W resident in LDS, two rows per workgroup (one workgroup per CU: at training sizes -- 32 768 rows --
        // the streaming kernel's eight rows per workgroup and several workgroups per CU win, 1.5 vs 2.5 ms; round 3: also with the workgroup
        // walking over its row groups so that W is loaded once, 4.9 ms -- two rows per pass are two dependent fma chains per thread at one
        // wave per SIMD: latency-bound); rows per thread by how many row groups of n_out threads fit 256
        hipStream_t st = (hipStream_t)s;
        const int gh = 256 / Hn, gv = 256 / D;          // row groups available in the hidden / visible phase
        bool done = false;
#define TRY(R, RGH, RGV) (rbm_lds_resident_bytes(R, D, Hn) <= 158 * 1024 && \
                          launch_gibbs_lds<R, RGH, RGV>(st, N, D, Hn, k, v0, W, bh, ld_bh, bv, ld_bv, seed, row0, row_ids, sub0, p_v, v_out, seed_step))
        done = gv >= 2 ? (gh >= 2 ? TRY(2, 1, 1) : TRY(2, 2, 1)) : (gh >= 2 ? TRY(2, 1, 2) : TRY(2, 2, 2));
#undef TRY
        if (done) {
            MNN_LAUNCH_CHECK();
            return MNN_OK;
        }
    }
    if (gibbs_mfma_lds_bytes(D, Hn) <= 158 * 1024 && getenv("MNN_RBM_NO_MFMA") == nullptr) {
        // training batches: the chain on the f32 matrix cores (same draws: see rbm_gibbs_mfma_kernel)
        static bool raised_[64];
        bool& raised = mnn_dev_flag(raised_);
        if (!raised) {
            MNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rbm_gibbs_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised = true;
        }
        GibbsMfmaArgs a{N, D, Hn, k, v0, W, bh, ld_bh, bv, ld_bv, seed, row0, row_ids, sub0, p_v, v_out, seed_step};
        hipLaunchKernelGGL(rbm_gibbs_mfma_kernel, dim3(cdiv(N, GM_ROWS)), dim3(512), gibbs_mfma_lds_bytes(D, Hn), (hipStream_t)s, a);
        MNN_LAUNCH_CHECK();
        return MNN_OK;
    }
    int rc = mnn_transpose(s, W, MNN_F32, D, Hn, Hn, workspace, MNN_F32, D);
    if (rc != MNN_OK) return rc;
    hipLaunchKernelGGL(rbm_gibbs_kernel, dim3(cdiv(N, RBM_R)), dim3(256), rbm_lds_bytes(D, Hn), (hipStream_t)s, N, D, Hn, k, v0, W,
                       (const float*)workspace, bh, ld_bh, bv, ld_bv, seed, row0, row_ids, sub0, p_v, v_out, seed_step);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_rbm_gibbs(mnn_stream_t s, int N, int D, int Hn, int k, const uint8_t* v0, const float* W, const float* bh, int ld_bh,
                             const float* bv, int ld_bv, uint64_t seed, uint32_t row0, const uint32_t* row_ids, uint32_t sub0, float* p_v,
                             uint8_t* v_out, void* workspace) {
    return mnn_rbm_gibbs_stepped(s, N, D, Hn, k, v0, W, bh, ld_bh, bv, ld_bv, seed, row0, row_ids, sub0, p_v, v_out, workspace, nullptr);
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
    rbm_phase(smem, Kp, K, Wk, ldw, n_out, b, ld_b, n0, N, [&](int r, int o, float z) {
        const int n = n0 + r;
        if (n >= N) return;
        const float p = det_sigmoid(z);
        if (p_out) p_out[(size_t)n * n_out + o] = p;
        if (s_out) {
            const float u = philox_uniform1(seed, (uint32_t)stream_id, row0 + (uint32_t)n, sub, (uint32_t)o);
            s_out[(size_t)n * n_out + o] = u < p ? 1 : 0;
        }
    });
}

// The half-step on the f32 matrix cores (same ascending fmaf chain per output: see rbm_gibbs_mfma_kernel): 64 rows per workgroup, the rows'
// inputs in LDS as f32 [64][odd pitch], Wk [K][n_out] in LDS, C[out unit][row] tiles of 32 x 32 spread over the 8 waves; a lane's accumulator
// quad = four consecutive outputs of one row = one Philox block.  From 2048 rows on (DBN encode / decode of a training batch, dbn.py:136-180,
// and the free-energy gradient's hidden passes).
template <typename TV>
__global__ void __launch_bounds__(512)
rbm_half_mfma_kernel(int N, int K, int n_out, const TV* __restrict__ in, const float* __restrict__ Wk, int ldw, const float* __restrict__ b, int ld_b,
                     int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* __restrict__ p_out, uint8_t* __restrict__ s_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ke = (K + 1) & ~1, lw = n_out | 1, pin = Ke | 1;
    float* Ws = smem;                       // [Ke][lw]
    float* xs = smem + (size_t)Ke * lw;     // [64][pin]
    float* tile_p = xs + (size_t)GM_ROWS * pin;                                   // [8 waves][32][33] f32: a job's probabilities, [row][unit]
    uint8_t* tile_s = reinterpret_cast<uint8_t*>(tile_p + 8 * 32 * 33);           // [8 waves][32][36] u8: its draws
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int k = w; k < Ke; k += 8)
        for (int o = lane; o < lw; o += 64) Ws[k * lw + o] = o < n_out ? Wk[(size_t)min(k, K - 1) * ldw + o] : 0.f;
    const int r = lane & 31, hh = lane >> 5;
    const int not_ = (n_out + 31) / 32;
    // persistent over 64-row tiles: the weights are fetched into LDS once per workgroup (one workgroup per CU), not once per 64 rows -- a
    // half-step has a single K-long chain per output, so a per-tile weight load (59 KB for 5.6 KB of inputs at 88 -> 168) was most of its time
    for (int tile = blockIdx.x; tile * GM_ROWS < N; tile += gridDim.x) {
        const int n0 = tile * GM_ROWS;
        __syncthreads();                                        // the previous tile's chains have read xs (first pass: Ws is complete)
        {   // the 64 rows' inputs: thread t -> row t >> 3, eight lanes walk its columns; sixteen loads in flight per thread (unconditional, clamped)
            const int rr = threadIdx.x >> 3, sub = threadIdx.x & 7, n = n0 + rr;
            const TV* __restrict__ src = in + (size_t)min(n, N - 1) * K;
            float* d = xs + rr * pin;
            for (int kb = sub; kb < pin; kb += 128) {
                TV v[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = src[min(kb + 8 * q, K - 1)];
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (kb + 8 * q < pin) d[kb + 8 * q] = (n < N && kb + 8 * q < K) ? (float)v[q] : 0.f;
            }
        }
        __syncthreads();
        for (int job = w; job < 2 * not_; job += 8) {
            const int rt = job / not_, ot = job - rt * not_;
            gm_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            const float* ap = Ws + (size_t)hh * lw + min(32 * ot + r, n_out - 1);
            const float* bp = xs + (32 * rt + r) * pin + hh;
            for (int s2 = 0; s2 < Ke / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(size_t)s2 * 2 * lw], bp[2 * s2], acc, 0, 0, 0);
            const int row = min(n0 + 32 * rt + r, N - 1);       // (rows past N repeat the last row; masked at the stores)
            // results through a wave-private LDS tile [row][unit], then out row-major: a lane holds four units of ONE row, so direct stores
            // would be 64 scattered 4-byte (and 1-byte) writes per instruction; from the tile every instruction writes two rows' 128-byte runs
            float* tp = tile_p + w * (32 * 33);
            uint8_t* ts = tile_s + w * (32 * 36);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int o0 = 32 * ot + 8 * g4 + 4 * hh;
                float u[4] = {0.f, 0.f, 0.f, 0.f};
                if (s_out) philox_uniform4(seed, (uint32_t)stream_id, row0 + (uint32_t)row, sub, (uint32_t)(o0 >> 2), u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = min(o0 + e, n_out - 1);
                    const float p = det_sigmoid(acc[4 * g4 + e] + b[(size_t)row * ld_b + o]);
                    tp[r * 33 + 8 * g4 + 4 * hh + e] = p;
                    ts[r * 36 + 8 * g4 + 4 * hh + e] = u[e] < p ? 1 : 0;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own LDS writes, then its reads (LDS serves a wave in order)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = i * 64 + lane, rr = idx >> 5, cc = idx & 31;
                const int row2 = n0 + 32 * rt + rr, o = 32 * ot + cc;
                if (row2 < N && o < n_out) {
                    if (p_out) p_out[(size_t)row2 * n_out + o] = tp[rr * 33 + cc];
                    if (s_out) s_out[(size_t)row2 * n_out + o] = ts[rr * 36 + cc];
                }
            }
        }
    }
}
static size_t half_mfma_lds_bytes(int K, int n_out) {
    const int Ke = (K + 1) & ~1;
    return ((size_t)Ke * (n_out | 1) + (size_t)GM_ROWS * (Ke | 1) + 8 * 32 * 33) * sizeof(float) + 8 * 32 * 36;
}

static int launch_half(hipStream_t st, int N, int K, int n_out, const void* in, int in_dtype, const float* Wk, int ldw, const float* b,
                       int ld_b, int stream_id, uint64_t seed, uint32_t row0, uint32_t sub, float* p_out, uint8_t* s_out) {
    // matrix-core form for SAMPLING half-steps of training batches (DBN encode / decode: its accumulator layout gives every Philox block to one
    // lane); a probabilities-only pass (the free-energy gradient's hidden activations) stays on the vector kernel, whose stores are
    // coalesced -- measured at [32768, 88 -> 256]: 0.11 ms vector vs 0.19 ms matrix-core per call
    if (N >= 2048 && s_out != nullptr && half_mfma_lds_bytes(K, n_out) <= 158 * 1024 && getenv("MNN_RBM_NO_MFMA") == nullptr) {
        static bool raised_[64];
        bool& raised = mnn_dev_flag(raised_);
        if (!raised) {
            MNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rbm_half_mfma_kernel<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            MNN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rbm_half_mfma_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            raised = true;
        }
        const size_t l2 = half_mfma_lds_bytes(K, n_out);
        int dev = 0, cus = 256;
        MNN_HIP(hipGetDevice(&dev));
        MNN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        const int grid = min(cdiv(N, GM_ROWS), max(cus, 1));        // one workgroup per CU walks the row tiles
        if (in_dtype == MNN_U8)
            hipLaunchKernelGGL(rbm_half_mfma_kernel<uint8_t>, dim3(grid), dim3(512), l2, st, N, K, n_out, (const uint8_t*)in, Wk, ldw, b, ld_b,
                               stream_id, seed, row0, sub, p_out, s_out);
        else
            hipLaunchKernelGGL(rbm_half_mfma_kernel<float>, dim3(grid), dim3(512), l2, st, N, K, n_out, (const float*)in, Wk, ldw, b, ld_b,
                               stream_id, seed, row0, sub, p_out, s_out);
        MNN_LAUNCH_CHECK();
        return MNN_OK;
    }
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
                       const float* __restrict__ bv, int ld_bv, float* __restrict__ F, float* __restrict__ p_h) {
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
    rbm_phase(smem, Dp, D, W, Hn, Hn, bh, ld_bh, n0, N, [&](int r, int j, float z) {
        const int n = n0 + r;
        if (n < N) {
            part[r] -= softplus_f(z);
            // d F / d z = -sigmoid(z): the backward pass's hidden activations, from the pre-activation this pass has anyway (the same
            // det_sigmoid as mnn_rbm_hidden, so the gradient keeps its bits; saves that pass re-reading v and W and re-forming z)
            if (p_h != nullptr) p_h[(size_t)n * Hn + j] = det_sigmoid(z);
        }
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
                                   const float* bv, int ld_bv, float* F, float* p_h) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && v && W && bh && bv && F, "mnn_rbm_free_energy: bad arguments");
    MNN_REQUIRE((ld_bh == 0 || ld_bh >= Hn) && (ld_bv == 0 || ld_bv >= D), "mnn_rbm_free_energy: bad bias leading dimension");
    const size_t lds = (size_t)RBM_R * ((D + 3) & ~3) * sizeof(float);
    hipLaunchKernelGGL(rbm_free_energy_kernel, dim3(cdiv(N, RBM_R)), dim3(256), lds, (hipStream_t)s, N, D, Hn, v, W, bh, ld_bh, bv, ld_bv, F, p_h);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// CD-k bias deltas (rbm.py:318-327):  dbv[d] += scale * sum_n (v[n,d] - p_v[n,d]),  dbh[j] += scale * sum_n (h[n,j] - p_h[n,j]).
// One pass over the four [N, .] arrays; a workgroup owns 64 columns of one of the two outputs and a slab of rows, threads of a
// wave read consecutive columns (coalesced), the 4 waves split the slab's rows; one f32 atomic per column and workgroup.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rbm_cd_bias_delta_kernel(int N, int D, int Hn, const uint8_t* __restrict__ v, const float* __restrict__ p_v, const uint8_t* __restrict__ h,
                         const float* __restrict__ p_h, float scale, float* __restrict__ dbv, float* __restrict__ dbh) {
    __shared__ float part[4][64];
    const int nbv = (D + 63) / 64;
    const bool vis = (int)blockIdx.x < nbv;
    const int cols = vis ? D : Hn;
    const int c = (vis ? blockIdx.x : blockIdx.x - nbv) * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const uint8_t* s = vis ? v : h;
    const float* p = vis ? p_v : p_h;
    float acc = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * 4 + w; r < N; r += gridDim.y * 4) acc += (float)s[(size_t)r * cols + c] - p[(size_t)r * cols + c];
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < cols) atomicAdd((vis ? dbv : dbh) + c, scale * (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]));
}

extern "C" int mnn_rbm_cd_bias_delta(mnn_stream_t s, int N, int D, int Hn, const uint8_t* v, const float* p_v, const uint8_t* h, const float* p_h,
                                     float scale, float* dbv, float* dbh) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && v && p_v && h && p_h && dbv && dbh, "mnn_rbm_cd_bias_delta: bad arguments");
    const int slabs = N >= 4096 ? 64 : (N >= 256 ? 8 : 1);
    hipLaunchKernelGGL(rbm_cd_bias_delta_kernel, dim3((D + 63) / 64 + (Hn + 63) / 64, slabs), dim3(256), 0, (hipStream_t)s, N, D, Hn, v, p_v, h, p_h,
                       scale, dbv, dbh);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// out[i] = a * x[i] + b * y[i]  (f32; out may alias x or y; y may be NULL when b == 0): the `assign_add` of the CD deltas
// (rbm.py:329-333) and the combination of their positive / negative phase products.
__global__ void __launch_bounds__(256) axpby_kernel(long n, float a, const float* x, float b, const float* y, float* out) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
extern "C" int mnn_axpby_f32(mnn_stream_t s, long n, float a, const float* x, float b, const float* y, float* out) {
    MNN_REQUIRE(n > 0 && x && out && (y || b == 0.f), "mnn_axpby_f32: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3((int)min(1024L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, n, a, x, b, y, out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// rbm.py:286-297: bv[d] = log(1e-6 + p/(1-p)) with p = colsum[d] / count (utils/auxiliary.py:9-11 safe_log); colsum comes from
// mnn_bias_grad over the f32 batch (summed over ranks by the caller under data parallelism).
__global__ void __launch_bounds__(256) rbm_visible_bias_init_kernel(int D, const float* __restrict__ colsum, float count, float* __restrict__ bv) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d < D) {
        const float p = colsum[d] / count;
        bv[d] = logf(1e-6f + p / (1.f - p));
    }
}
extern "C" int mnn_rbm_visible_bias_init(mnn_stream_t s, int D, const float* colsum, float count, float* bv) {
    MNN_REQUIRE(D > 0 && colsum && bv && count > 0.f, "mnn_rbm_visible_bias_init: bad arguments");
    hipLaunchKernelGGL(rbm_visible_bias_init_kernel, dim3(cdiv(D, 256)), dim3(256), 0, (hipStream_t)s, D, colsum, count, bv);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// Rows of the LSTM-RBM cost gradient (rnn_rbm.py:113-126 through rbm.py:229: cost = F(v) - F(v_s), v_s constant):
//   d cost / d bh_t = w (sigmoid(z(v_s)) - sigmoid(z(v))),   d cost / d bv_t = w (v_s - v),   w = row_weight * scale
// written into the Dense-output-shaped block d_out [N, ld] (columns [Hn + D, ld) zeroed), together with the two scaled hidden blocks whose
// products with v_s^T and v^T give d cost / d W = v_s^T (w ss) - v^T (w sv):  pos = w ss,  neg = -w sv  (so that both GEMMs ACCUMULATE).
__global__ void __launch_bounds__(256) rbm_cd_rows_kernel(int N, int D, int Hn, int ld, const uint8_t* __restrict__ v, const uint8_t* __restrict__ vs,
                                                          const float* __restrict__ sv, const float* __restrict__ ss, const float* __restrict__ rw,
                                                          float scale, float* __restrict__ d_out, float* __restrict__ pos, float* __restrict__ neg) {
    const int n = blockIdx.x;
    const float w = rw[n] * scale;
    for (int j = threadIdx.x; j < ld; j += 256) {
        float o = 0.f;
        if (j < Hn) {
            const float a = ss[(size_t)n * Hn + j], b = sv[(size_t)n * Hn + j];
            o = w * (a - b);
            pos[(size_t)n * Hn + j] = w * a;
            neg[(size_t)n * Hn + j] = -(w * b);
        } else if (j < Hn + D) {
            const int i = j - Hn;
            o = w * ((float)vs[(size_t)n * D + i] - (float)v[(size_t)n * D + i]);
        }
        d_out[(size_t)n * ld + j] = o;
    }
}
extern "C" int mnn_rbm_cd_rows(mnn_stream_t s, int N, int D, int Hn, int ld, const uint8_t* v, const uint8_t* v_s, const float* sv, const float* ss,
                               const float* row_weight, float scale, float* d_out, float* pos, float* neg) {
    MNN_REQUIRE(N > 0 && D > 0 && Hn > 0 && ld >= Hn + D && v && v_s && sv && ss && row_weight && d_out && pos && neg, "mnn_rbm_cd_rows: bad arguments");
    hipLaunchKernelGGL(rbm_cd_rows_kernel, dim3(N), dim3(256), 0, (hipStream_t)s, N, D, Hn, ld, v, v_s, sv, ss, row_weight, scale, d_out, pos, neg);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// dz[i] = dy[i] * y[i] * (1 - y[i])  (f32; dz may alias dy): backward of y = sigmoid(z), the activation of the Dense feedback module
// (dnn.py:60-76 / multinn_feedback.py:48-52)
__global__ void __launch_bounds__(256) sigmoid_grad_kernel(long n, const float* __restrict__ dy, const float* __restrict__ y, float* dz) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float yy = y[i];
        dz[i] = dy[i] * (yy - yy * yy);
    }
}
extern "C" int mnn_sigmoid_grad_f32(mnn_stream_t s, long n, const float* dy, const float* y, float* dz) {
    MNN_REQUIRE(n > 0 && dy && y && dz, "mnn_sigmoid_grad_f32: bad arguments");
    hipLaunchKernelGGL(sigmoid_grad_kernel, dim3((int)min(2048L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, n, dy, y, dz);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
