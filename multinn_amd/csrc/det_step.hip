// Deterministic f32 single steps of the sampling scan (rnn_estimator.py:293-323: sample_single -> single_step; rnn_nade.py:253-277;
// multinn_feedback.py:175-218): one LSTMBlockCell step and one Dense layer whose every output is a FIXED sequence of IEEE operations, so
// that a CPU restatement (oracle/det_ref.c) reproduces every bit and `generate()` can be checked draw by draw over a whole scan
// ("bit-exact for Bernoulli sampling indices under a fixed RNG", BASELINE.json) -- the throughput kernels use the hardware's exp2 / rcp
// approximations, whose results no C program can restate.
//
// Specification (DESIGN.md "Deterministic sampling"; oracle/det_ref.c restates it independently):
//   z[col]  = fmaf-chain over k ascending of xh[k] * W[k][col], starting from 0, xh = [x | x2 | h_prev] (TF's concat order, rnn.py:124
//             LSTMBlockCell xh = [x, h]);  then + bias[col]            (one rounding per product-add, one for the bias add)
//   i, f, o = det_sigmoid(z)   ci = det_tanh(z) = 2 det_sigmoid(2 z) - 1
//   c       = round(ci * i) + round(c_prev * f)      h = det_tanh(c) * o        (forget_bias 0, no peephole, no clipping)
//   Dense:  out[n] = fmaf-chain over k ascending of x[k] * W[k][n] from 0; then + bias[n]
// Master weights are read in their TF layout and in f32 (no packed / 16-bit copies): sampling runs in the reference's own arithmetic
// whatever the training precision is.
//
// Mapping (sampling batches are small -- 72 rows by default, default_config.yaml:43-51): a workgroup = 64 units x 4 gates (thread = one
// pre-activation column, so a wave reads 256 contiguous bytes of a W row) x R rows; the R rows of xh sit in LDS and are broadcast;
// each thread carries R independent chains (the chain of ONE output is sequential by definition, 4 cycles per link and wave).  The four
// gates of a unit meet through LDS for the pointwise part.  Several (generator, layer) jobs -- the M per-track generators of the feedback
// scan -- run as ONE launch (blockIdx.z = job).
#include "common.h"

#define DS_R 6                       // rows per workgroup
#define DS_MAXJOBS MNN_DET_MAX_JOBS

struct DetLstmJobs { mnn_det_lstm_job job[DS_MAXJOBS]; };
struct DetDenseJobs { mnn_det_dense_job job[DS_MAXJOBS]; };

__device__ __forceinline__ float det_tanh(float x) { return __fsub_rn(__fmul_rn(2.0f, det_sigmoid(__fmul_rn(2.0f, x))), 1.0f); }

// xh[r][k] of the job for rows r0 .. r0 + R - 1 into LDS (rows past B: zeros), k-contiguous, pitch Kp (multiple of 4)
__device__ __forceinline__ void ds_stage_rows(const mnn_det_lstm_job& jb, int B, int r0, int K, int Kp, float* xs) {
    const int n1 = jb.n_x, n2 = jb.n_x2, u = jb.units;
    for (int e = threadIdx.x; e < DS_R * Kp; e += blockDim.x) {
        const int r = e / Kp, k = e - r * Kp, row = r0 + r;
        float v = 0.f;
        if (row < B && k < K) {
            if (k < n1) v = jb.x_dtype == MNN_U8 ? (float)reinterpret_cast<const uint8_t*>(jb.x)[(size_t)row * jb.ld_x + k]
                                                  : reinterpret_cast<const float*>(jb.x)[(size_t)row * jb.ld_x + k];
            else if (k < n1 + n2) v = jb.x2[(size_t)row * jb.ld_x2 + (k - n1)];
            else if (jb.h_prev != nullptr) v = jb.h_prev[(size_t)row * u + (k - n1 - n2)];
        }
        xs[e] = v;
    }
}

// R chains of one column: acc[r] = fma(xs[r][k], W[k][col], acc[r]), k ascending.  Eight W rows are requested ahead of their FMAs.
__device__ __forceinline__ void ds_chains(const float* __restrict__ w, size_t ldw, int K, const float* xs, int Kp, float (&acc)[DS_R]) {
#pragma unroll
    for (int r = 0; r < DS_R; ++r) acc[r] = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = w[(size_t)(k + j) * ldw];
#pragma unroll
        for (int r = 0; r < DS_R; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(xs + r * Kp + k);
            const float4 b = *reinterpret_cast<const float4*>(xs + r * Kp + k + 4);
            float t = acc[r];
            t = fmaf(a.x, wv[0], t); t = fmaf(a.y, wv[1], t); t = fmaf(a.z, wv[2], t); t = fmaf(a.w, wv[3], t);
            t = fmaf(b.x, wv[4], t); t = fmaf(b.y, wv[5], t); t = fmaf(b.z, wv[6], t); t = fmaf(b.w, wv[7], t);
            acc[r] = t;
        }
    }
    for (; k < K; ++k) {
        const float wv = w[(size_t)k * ldw];
#pragma unroll
        for (int r = 0; r < DS_R; ++r) acc[r] = fmaf(xs[r * Kp + k], wv, acc[r]);
    }
}

__global__ void __launch_bounds__(256) lstm_step_det_kernel(DetLstmJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const mnn_det_lstm_job& jb = J.job[blockIdx.z];
    const int u = jb.units, ub = blockIdx.x * 64;
    if (ub >= u) return;                             // the grid covers the widest job
    const int r0 = blockIdx.y * DS_R;
    const int K = jb.n_x + jb.n_x2 + u, Kp = (K + 3) & ~3;
    float* xs = ds_smem;                             // [R][Kp]
    float* zs = ds_smem + DS_R * Kp;                 // [4][R][64]
    ds_stage_rows(jb, B, r0, K, Kp, xs);
    __syncthreads();
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int unit = min(ub + l, u - 1);             // units are a multiple of 32: the upper half of the last workgroup may repeat a column
    const int col = g * u + unit;
    float acc[DS_R];
    ds_chains(jb.W + col, (size_t)4 * u, K, xs, Kp, acc);
    const float bv = jb.bias[col];
#pragma unroll
    for (int r = 0; r < DS_R; ++r) zs[(g * DS_R + r) * 64 + l] = __fadd_rn(acc[r], bv);
    __syncthreads();
    for (int e = threadIdx.x; e < DS_R * 64; e += 256) {
        const int r = e >> 6, ll = e & 63, row = r0 + r, un = ub + ll;
        if (row >= B || un >= u) continue;
        const float gi = det_sigmoid(zs[(0 * DS_R + r) * 64 + ll]), gc = det_tanh(zs[(1 * DS_R + r) * 64 + ll]);
        const float gf = det_sigmoid(zs[(2 * DS_R + r) * 64 + ll]), go = det_sigmoid(zs[(3 * DS_R + r) * 64 + ll]);
        const float cp = jb.c_prev != nullptr ? jb.c_prev[(size_t)row * u + un] : 0.f;
        const float c = __fadd_rn(__fmul_rn(gc, gi), __fmul_rn(cp, gf));
        jb.c_out[(size_t)row * u + un] = c;
        jb.h_out[(size_t)row * u + un] = __fmul_rn(det_tanh(c), go);
    }
}

__global__ void __launch_bounds__(256) dense_det_kernel(DetDenseJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const mnn_det_dense_job& jb = J.job[blockIdx.z];
    const int nb = blockIdx.x * 256;
    if (nb >= jb.N) return;
    const int r0 = blockIdx.y * DS_R;
    const int K = jb.K, Kp = (K + 3) & ~3;
    float* xs = ds_smem;
    for (int e = threadIdx.x; e < DS_R * Kp; e += 256) {
        const int r = e / Kp, k = e - r * Kp, row = r0 + r;
        xs[e] = (row < B && k < K) ? jb.x[(size_t)row * jb.ld_x + k] : 0.f;
    }
    __syncthreads();
    const int n = min(nb + (int)threadIdx.x, jb.N - 1);
    float acc[DS_R];
    ds_chains(jb.W + n, (size_t)jb.ld_w, K, xs, Kp, acc);
    const float bv = jb.bias != nullptr ? jb.bias[n] : 0.f;
    if (nb + (int)threadIdx.x < jb.N) {
#pragma unroll
        for (int r = 0; r < DS_R; ++r)
            if (r0 + r < B) jb.out[(size_t)(r0 + r) * jb.ld_out + n] = __fadd_rn(acc[r], bv);
    }
}

extern "C" int mnn_lstm_step_det(mnn_stream_t s, int B, int njobs, const mnn_det_lstm_job* jobs) {
    MNN_REQUIRE(B > 0 && njobs > 0 && njobs <= DS_MAXJOBS && jobs != nullptr, "mnn_lstm_step_det: 1..%d jobs, B > 0", DS_MAXJOBS);
    DetLstmJobs J;
    memset(&J, 0, sizeof(J));
    int umax = 0, kmax = 0;
    for (int j = 0; j < njobs; ++j) {
        const mnn_det_lstm_job& jb = jobs[j];
        MNN_REQUIRE(jb.units > 0 && jb.units % 32 == 0 && jb.W && jb.bias && jb.c_out && jb.h_out, "mnn_lstm_step_det: job %d: units %% 32, W, bias, c_out, h_out", j);
        MNN_REQUIRE(jb.n_x >= 0 && jb.n_x2 >= 0 && (jb.n_x == 0 || (jb.x && jb.ld_x >= jb.n_x && (jb.x_dtype == MNN_U8 || jb.x_dtype == MNN_F32))) &&
                    (jb.n_x2 == 0 || (jb.x2 && jb.ld_x2 >= jb.n_x2)), "mnn_lstm_step_det: job %d: input blocks", j);
        MNN_REQUIRE((jb.h_prev == nullptr) == (jb.c_prev == nullptr), "mnn_lstm_step_det: job %d: h_prev and c_prev come together", j);
        J.job[j] = jb;
        umax = max(umax, jb.units);
        kmax = max(kmax, jb.n_x + jb.n_x2 + jb.units);
    }
    const size_t lds = ((size_t)DS_R * ((kmax + 3) & ~3) + 4 * DS_R * 64) * sizeof(float);
    MNN_REQUIRE(lds <= 64 * 1024, "mnn_lstm_step_det: %d inputs + units do not fit the staging buffer", kmax);
    dim3 grid(cdiv(umax, 64), cdiv(B, DS_R), njobs);
    hipLaunchKernelGGL(lstm_step_det_kernel, grid, dim3(256), lds, (hipStream_t)s, J, B);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_dense_det(mnn_stream_t s, int B, int njobs, const mnn_det_dense_job* jobs) {
    MNN_REQUIRE(B > 0 && njobs > 0 && njobs <= DS_MAXJOBS && jobs != nullptr, "mnn_dense_det: 1..%d jobs, B > 0", DS_MAXJOBS);
    DetDenseJobs J;
    memset(&J, 0, sizeof(J));
    int nmax = 0, kmax = 0;
    for (int j = 0; j < njobs; ++j) {
        const mnn_det_dense_job& jb = jobs[j];
        MNN_REQUIRE(jb.K > 0 && jb.N > 0 && jb.x && jb.W && jb.out && jb.ld_x >= jb.K && jb.ld_w >= jb.N && jb.ld_out >= jb.N,
                    "mnn_dense_det: job %d: K, N > 0, x, W, out, leading dimensions", j);
        J.job[j] = jb;
        nmax = max(nmax, jb.N);
        kmax = max(kmax, jb.K);
    }
    const size_t lds = (size_t)DS_R * ((kmax + 3) & ~3) * sizeof(float);
    MNN_REQUIRE(lds <= 64 * 1024, "mnn_dense_det: K = %d does not fit the staging buffer", kmax);
    dim3 grid(cdiv(nmax, 256), cdiv(B, DS_R), njobs);
    hipLaunchKernelGGL(dense_det_kernel, grid, dim3(256), lds, (hipStream_t)s, J, B);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
