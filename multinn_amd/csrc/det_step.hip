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
// Mapping (sampling batches are small -- 72 rows by default, default_config.yaml:43-51): the chains run on the f32 MATRIX cores.
// v_mfma_f32_32x32x2_f32 computes D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)), one IEEE rounding per product-add (cdna_hip_programming.md,
// "FP32-input MFMA"), so a run of them over ascending k IS the specified fmaf chain -- 1024 chains per wave and instruction instead of one
// per lane.  (Round 4's first form, a vector fma chain per thread with the rows' inputs broadcast from LDS, took ~170 us per step of the
// [512, 256] stack at 72 rows: every group of eight weight loads was waited for at L2 latency.)  The product is formed transposed,
// C[column][row] = sum_k W[k][column] xh[row][k] (A = weights straight from global memory, a ring of PF k-pairs ahead of their MFMAs;
// B = the 32 rows' inputs from LDS, f32 [32][odd pitch]: conflict-free).  LSTM: a wave's 32 A rows are the FOUR gates of 8 units
// (row g * 8 + uu <-> TF column g * units + unit), so a lane's accumulator quads hold i, ci, f, o of four (row, unit) pairs and the
// pointwise part needs no exchange.  A workgroup = 4 waves = 32 units x 32 rows; K is walked in chunks of DS_KC through the staging
// buffer.  Several (generator, layer) jobs -- the M per-track generators of the feedback scan -- run as ONE launch (blockIdx.z = job).
#include "common.h"

#define DS_KC 1024                   // k per staging chunk (f32 [32][DS_KC + 1] = 128 KiB)
#define DS_PF 28                     // k-pairs per weight register set; two sets alternate, so every load has 28 .. 56 MFMAs (0.75 .. 1.5 us) to land
#define DS_MAXJOBS MNN_DET_MAX_JOBS
typedef float ds_f32x16 __attribute__((ext_vector_type(16)));

struct DetLstmJobs { mnn_det_lstm_job job[DS_MAXJOBS]; };
struct DetDenseJobs { mnn_det_dense_job job[DS_MAXJOBS]; };

__device__ __forceinline__ float det_tanh(float x) { return __fsub_rn(__fmul_rn(2.0f, det_sigmoid(__fmul_rn(2.0f, x))), 1.0f); }

// One chunk of a wave's chain: acc += sum over k in [k0, k0 + kc) (kc even, zero-padded inputs) of W[k][col] * xs[row][k - k0].
// w_k(k) returns the lane's weight of row k (clamped to the last row: its input is zero there); loads run DS_PF k-pairs ahead.
template <typename WF>
__device__ __forceinline__ void ds_chain_chunk(WF&& w_k, int k0, int kc, int K, const float* __restrict__ xs_row, int hh, ds_f32x16& acc) {
    // Two register sets of DS_PF weights alternate: while the MFMAs of one set run, the loads of the set after next are in flight, and the wait
    // in front of a set only covers loads issued a whole set earlier.  (A single ring refilled slot by slot compiles to `s_waitcnt vmcnt(0)` at
    // the loop header -- the last refills are then waited for at full Infinity-Cache latency once per round: ~38 us per launch at K = 952
    // with 16 slots, where the chain itself is 13.)  Loads are unconditional from clamped rows (the inputs are zero there).
    const int np = kc / 2;
    float wa[2][DS_PF];
#pragma unroll
    for (int j = 0; j < DS_PF; ++j) wa[0][j] = w_k(min(k0 + 2 * j + hh, K - 1));
    for (int s0 = 0; s0 < np; s0 += 2 * DS_PF) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int sb = s0 + half * DS_PF;                 // first k-pair of the set about to be consumed
#pragma unroll
            for (int j = 0; j < DS_PF; ++j) wa[half ^ 1][j] = w_k(min(k0 + 2 * (sb + DS_PF + j) + hh, K - 1));
            // ... and this set's input operands out of LDS, all of them in front of the first MFMA: with a read + s_waitcnt lgkmcnt(0) in front
            // of every MFMA the 64-cycle instruction waited another ~64 cycles for its operand each time
            float xb[DS_PF];
#pragma unroll
            for (int j = 0; j < DS_PF; ++j) xb[j] = xs_row[2 * min(sb + j, np - 1) + hh];
            __builtin_amdgcn_sched_barrier(0);                // the refill loads and the LDS reads first, then this set's MFMAs
            if (sb < np) {
#pragma unroll
                for (int j = 0; j < DS_PF; ++j)
                    if (sb + j < np) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[half][j], xb[j], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Staging: one SEGMENT (a run of columns of one source array) of the 32 rows into LDS, f32 [32][pitch] at column offset `c0`; rows past B
// and columns outside the chunk are skipped / zero.  Eight loads per thread are in flight before the first is used (the first version staged
// element by element behind a three-way source test: every element waited for its own memory round trip, ~100 us per step).
// Thread t: row t >> 3, lanes t & 7 walk the columns; loads are unconditional from clamped addresses.
template <typename T>
__device__ __forceinline__ void ds_stage_seg(const T* __restrict__ src, size_t ld, size_t es, int n, int seg0, int B, int r0, int k0, int kc,
                                             int pitch, float* xs) {
    // columns of this segment that fall into the chunk [k0, k0 + kc): global k = seg0 + j, j in [j_lo, j_hi)
    const int j_lo = max(0, k0 - seg0), j_hi = min(n, k0 + kc - seg0);
    if (j_lo >= j_hi) return;
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const bool rv = r0 + r < B;
    const T* __restrict__ p = src + (size_t)min(r0 + r, B - 1) * ld;
    float* d = xs + r * pitch + (seg0 - k0);
    typedef T T4 __attribute__((ext_vector_type(4)));
    if (es == 1 && ((uintptr_t)src % (4 * sizeof(T))) == 0 && ld % 4 == 0 && j_lo % 4 == 0 && (j_hi - j_lo) % 4 == 0) {
        // contiguous, aligned: four elements per load, up to sixteen loads in flight per thread (512 f32 per row = one batch)
        const int nv = (j_hi - j_lo) / 4;
        const T4* __restrict__ p4 = reinterpret_cast<const T4*>(p + j_lo);
        for (int v0 = sub; v0 < nv; v0 += 128) {
            T4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = p4[min(v0 + 8 * q, nv - 1)];
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (v0 + 8 * q < nv) {
                    float* dd = d + j_lo + 4 * (v0 + 8 * q);
                    dd[0] = rv ? (float)v[q].x : 0.f; dd[1] = rv ? (float)v[q].y : 0.f; dd[2] = rv ? (float)v[q].z : 0.f; dd[3] = rv ? (float)v[q].w : 0.f;
                }
        }
        return;
    }
    for (int j0 = j_lo + sub; j0 < j_hi; j0 += 128) {
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = p[(size_t)min(j0 + 8 * q, j_hi - 1) * es];
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (j0 + 8 * q < j_hi) d[j0 + 8 * q] = rv ? (float)v[q] : 0.f;
    }
}
// zero the columns [c_lo, c_hi) of the staged chunk (the even-K pad; a zero initial state)
__device__ __forceinline__ void ds_stage_zero(int c_lo, int c_hi, int pitch, float* xs) {
    const int r = threadIdx.x >> 3, sub = threadIdx.x & 7;
    for (int c = c_lo + sub; c < c_hi; c += 8) xs[r * pitch + c] = 0.f;
}

__global__ void __launch_bounds__(256) lstm_step_det_kernel(DetLstmJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const mnn_det_lstm_job& jb = J.job[blockIdx.z];
    const int u = jb.units, ub = blockIdx.x * 32;
    if (ub >= u) return;                             // the grid covers the widest job
    const int r0 = blockIdx.y * 32;
    const int n1 = jb.n_x, n2 = jb.n_x2;
    const int K = n1 + n2 + u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    // this lane's A row: gate g = r >> 3 of unit ub + 8 w + (r & 7)  ->  TF column g * u + unit
    const int col = (r >> 3) * u + ub + 8 * w + (r & 7);
    const float* __restrict__ wp = jb.W + col;
    const size_t ldw = (size_t)4 * u;
    auto w_k = [&](int k) { return wp[(size_t)k * ldw]; };
    ds_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += DS_KC) {
        const int kc = (min(DS_KC, K - k0) + 1) & ~1, pitch = kc | 1;
        if (k0 > 0) __syncthreads();
        if (n1 > 0) {
            if (jb.x_dtype == MNN_U8) ds_stage_seg(reinterpret_cast<const uint8_t*>(jb.x), (size_t)jb.ld_x, (size_t)jb.es_x, n1, 0, B, r0, k0, kc, pitch, ds_smem);
            else ds_stage_seg(reinterpret_cast<const float*>(jb.x), (size_t)jb.ld_x, (size_t)jb.es_x, n1, 0, B, r0, k0, kc, pitch, ds_smem);
        }
        if (n2 > 0) ds_stage_seg(jb.x2, (size_t)jb.ld_x2, (size_t)1, n2, n1, B, r0, k0, kc, pitch, ds_smem);
        if (jb.h_prev != nullptr) ds_stage_seg(jb.h_prev, (size_t)u, (size_t)1, u, n1 + n2, B, r0, k0, kc, pitch, ds_smem);
        else ds_stage_zero(max(0, n1 + n2 - k0), min(kc, K - k0), pitch, ds_smem);
        if (K - k0 < kc) ds_stage_zero(K - k0, kc, pitch, ds_smem);          // the pad column of an odd K
        __syncthreads();
        ds_chain_chunk(w_k, k0, kc, K, ds_smem + r * pitch, hh, acc);
    }
    // accumulator register e: A row (e & 3) + 8 (e >> 2) + 4 hh = gate (e >> 2), unit offset (e & 3) + 4 hh; C column = batch row r
    const int row = r0 + r;
    if (row >= B) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int un = ub + 8 * w + 4 * hh + q;
        const float zi = __fadd_rn(acc[q], jb.bias[un]), zc = __fadd_rn(acc[4 + q], jb.bias[u + un]);
        const float zf = __fadd_rn(acc[8 + q], jb.bias[2 * u + un]), zo = __fadd_rn(acc[12 + q], jb.bias[3 * u + un]);
        const float gi = det_sigmoid(zi), gc = det_tanh(zc), gf = det_sigmoid(zf), go = det_sigmoid(zo);
        const float cp = jb.c_prev != nullptr ? jb.c_prev[(size_t)row * u + un] : 0.f;
        const float c = __fadd_rn(__fmul_rn(gc, gi), __fmul_rn(cp, gf));
        jb.c_out[(size_t)row * u + un] = c;
        jb.h_out[(size_t)row * u + un] = __fmul_rn(det_tanh(c), go);
    }
}

__global__ void __launch_bounds__(256) dense_det_kernel(DetDenseJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const mnn_det_dense_job& jb = J.job[blockIdx.z];
    const int nb = blockIdx.x * 128;
    if (nb >= jb.N) return;
    const int r0 = blockIdx.y * 32;
    const int K = jb.K;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int n = min(nb + 32 * w + r, jb.N - 1);    // this lane's A row = output column n
    const float* __restrict__ wp = jb.W + n;
    const size_t ldw = (size_t)jb.ld_w;
    auto w_k = [&](int k) { return wp[(size_t)k * ldw]; };
    ds_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += DS_KC) {
        const int kc = (min(DS_KC, K - k0) + 1) & ~1, pitch = kc | 1;
        if (k0 > 0) __syncthreads();
        ds_stage_seg(jb.x, (size_t)jb.ld_x, (size_t)1, K, 0, B, r0, k0, kc, pitch, ds_smem);
        if (K - k0 < kc) ds_stage_zero(K - k0, kc, pitch, ds_smem);
        __syncthreads();
        ds_chain_chunk(w_k, k0, kc, K, ds_smem + r * pitch, hh, acc);
    }
    const int row = r0 + r;
    if (row >= B) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int nn = nb + 32 * w + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (nn < jb.N) jb.out[(size_t)row * jb.ld_out + nn] = jb.bias != nullptr ? __fadd_rn(acc[e], jb.bias[nn]) : acc[e];
    }
}

static size_t ds_lds_bytes(int K) { const int kc = (min(DS_KC, K) + 1) & ~1; return (size_t)32 * (kc | 1) * sizeof(float); }
static hipError_t ds_raise_lds() {
    static bool raised = false;
    if (raised) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_step_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = e == hipSuccess;
    return e;
}

extern "C" int mnn_lstm_step_det(mnn_stream_t s, int B, int njobs, const mnn_det_lstm_job* jobs) {
    MNN_REQUIRE(B > 0 && njobs > 0 && njobs <= DS_MAXJOBS && jobs != nullptr, "mnn_lstm_step_det: 1..%d jobs, B > 0", DS_MAXJOBS);
    DetLstmJobs J;
    memset(&J, 0, sizeof(J));
    int umax = 0, kmax = 0;
    for (int j = 0; j < njobs; ++j) {
        const mnn_det_lstm_job& jb = jobs[j];
        MNN_REQUIRE(jb.units > 0 && jb.units % 32 == 0 && jb.W && jb.bias && jb.c_out && jb.h_out, "mnn_lstm_step_det: job %d: units %% 32, W, bias, c_out, h_out", j);
        MNN_REQUIRE(jb.n_x >= 0 && jb.n_x2 >= 0 && (jb.n_x == 0 || (jb.x && jb.es_x >= 1 && (jb.x_dtype == MNN_U8 || jb.x_dtype == MNN_F32))) &&
                    (jb.n_x2 == 0 || (jb.x2 && jb.ld_x2 >= jb.n_x2)), "mnn_lstm_step_det: job %d: input blocks", j);
        MNN_REQUIRE((jb.h_prev == nullptr) == (jb.c_prev == nullptr), "mnn_lstm_step_det: job %d: h_prev and c_prev come together", j);
        J.job[j] = jb;
        umax = max(umax, jb.units);
        kmax = max(kmax, jb.n_x + jb.n_x2 + jb.units);
    }
    MNN_HIP(ds_raise_lds());
    dim3 grid(umax / 32, cdiv(B, 32), njobs);
    hipLaunchKernelGGL(lstm_step_det_kernel, grid, dim3(256), ds_lds_bytes(kmax), (hipStream_t)s, J, B);
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
    MNN_HIP(ds_raise_lds());
    dim3 grid(cdiv(nmax, 128), cdiv(B, 32), njobs);
    hipLaunchKernelGGL(dense_det_kernel, grid, dim3(256), ds_lds_bytes(kmax), (hipStream_t)s, J, B);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
