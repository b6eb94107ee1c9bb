// Deterministic f32 single steps of the sampling scan (rnn_estimator.py:293-323: sample_single -> single_step; rnn_nade.py:253-277;
// multinn_feedback.py:175-218): one LSTMBlockCell step and one Dense layer whose every output is a FIXED sequence of IEEE operations, so
// that a CPU restatement (oracle/det_ref.c) reproduces every bit and `generate()` can be checked draw by draw over a whole scan
// ("bit-exact for Bernoulli sampling indices under a fixed RNG", BASELINE.json) -- the throughput kernels use the hardware's exp2 / rcp
// approximations, whose results no C program can restate.
//
// Specification (DESIGN.md "Deterministic sampling"; oracle/det_ref.c restates it independently):
//   z[col]  = ((p_0 + p_1) + (p_2 + p_3)) + bias[col], p_s = fmaf-chain over the k of quarter s ascending of xh[k] * W[k][col], starting from 0;
//             xh = [x | x2 | h_prev] (TF's concat order, rnn.py:124 LSTMBlockCell xh = [x, h]); quarter s = k-pairs [s q, (s + 1) q) of
//             the K (rounded up to even) columns, q = ceil(pairs / 4)  (chunks of DS_KC = 1024 columns are quartered one by one)
//   i, f, o = det_sigmoid(z)   ci = det_tanh(z) = 2 det_sigmoid(2 z) - 1
//   c       = round(ci * i) + round(c_prev * f)      h = det_tanh(c) * o        (forget_bias 0, no peephole, no clipping)
//   Dense:  out[n] = ((p_0 + p_1) + (p_2 + p_3)) + bias[n], the same four quarter chains over x[k] * W[k][n]
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
// pointwise part needs no exchange of gates.  A workgroup = 4 waves = the 4 K quarters of 8 units x 32 rows (partial sums meet in LDS); K is
// staged in chunks of DS_KC.  Several (generator, layer) jobs -- the M per-track generators of the feedback scan -- run as ONE launch (blockIdx.z = job).
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
            if (sb + DS_PF <= np) {                           // a full set: 28 MFMAs in a straight line (a test per MFMA put each one behind two jumps)
#pragma unroll
                for (int j = 0; j < DS_PF; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[half][j], xb[j], acc, 0, 0, 0);
            } else if (sb < np) {                             // the chain's last, partial set
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

// K split (specification, restated by oracle/det_ref.c): the k-pairs of a staging chunk (all of K when K <= DS_KC) are cut into FOUR contiguous
// quarters of ceil(pairs / 4) pairs; quarter s has its own ascending fmaf chain p_s from 0 (running on over the chunks, if there are several),
// and  z = ((p_0 + p_1) + (p_2 + p_3)) + bias.  The four chains of an output run on the four waves of a workgroup at the same time: the
// sampling scan is latency-bound on a nearly idle chip (clocks drop to ~1 GHz there), and a K = 952 chain is 476 dependent 64-cycle MFMAs.
#define DS_SPLIT 4
__device__ __forceinline__ int ds_quarter(int kc) { return ((kc / 2) + DS_SPLIT - 1) / DS_SPLIT; }      // pairs per quarter of a chunk of kc (even) columns

// ---- packed weights (mnn_det_lstm_pack): the SAME numbers, laid out so that a wave's whole quarter chain is a handful of 16-byte loads ----
// The TF layout costs a wave one 4-byte load per MFMA (two 128-byte pieces 8 KB apart), at most 56 of them in flight: a K = 952 chain of 119
// MFMAs (64 cycles each: 3.2 us) took 9.6 us, every set of 28 waiting ~2.3 us for loads issued one set earlier
// (profiles/tools/det_step_trace.py).  Packed: Wp[((blk nchunks + chunk) 4 + quarter)][j4][lane = hh 32 + r][4] holds, for the lane that owns A row r
// (gate r >> 3 of unit 8 blk + (r & 7)) and k parity hh, the weights of k-pairs 4 j4 .. 4 j4 + 3 of that quarter (zero past its end).  A load
// instruction is then 1 KB contiguous, a K = 952 quarter is 30 of them, ALL issued before the inputs are even staged.
__host__ __device__ inline int ds_pack_q4max(int K) { const int kc = (min(DS_KC, K) + 1) & ~1; return (((kc / 2) + DS_SPLIT - 1) / DS_SPLIT + 3) / 4; }
__host__ __device__ inline int ds_pack_nchunks(int K) { return (K + DS_KC - 1) / DS_KC; }
#define DS_P4MAX 32                  // float4 per lane and quarter of a full chunk: ceil(ceil(512 / 4) / 4)

// dense_n == 0: LSTM kernel [K, 4 u] (block = 8 units x 4 gates); dense_n = N > 0: Dense kernel [K, ld_w >= N] (block = 32 columns, clamped at N - 1)
__global__ void __launch_bounds__(256) ds_pack_kernel(const float* __restrict__ W, int K, int u, float4* __restrict__ Wp, int dense_n, int ld_w) {
    const int q4max = ds_pack_q4max(K), nch = ds_pack_nchunks(K);
    const long total = (long)(dense_n > 0 ? (dense_n + 31) / 32 : u / 8) * nch * DS_SPLIT * q4max * 64;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long rest = t >> 6;
    const int j4 = (int)(rest % q4max); rest /= q4max;
    const int sq = (int)(rest % DS_SPLIT); rest /= DS_SPLIT;
    const int c = (int)(rest % nch);
    const int blk = (int)(rest / nch);
    const int r = lane & 31, hh = lane >> 5;
    const int k0 = c * DS_KC, kc = (min(DS_KC, K - k0) + 1) & ~1;
    const int q = ds_quarter(kc), p0 = min(sq * q, kc / 2), p1 = min(p0 + q, kc / 2);
    const size_t col = dense_n > 0 ? (size_t)min(blk * 32 + r, dense_n - 1) : (size_t)(r >> 3) * u + blk * 8 + (r & 7);
    const size_t ldw = dense_n > 0 ? (size_t)ld_w : (size_t)4 * u;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int pp = p0 + 4 * j4 + e, k = k0 + 2 * pp + hh;
        v[e] = (pp < p1 && k < K) ? W[(size_t)k * ldw + col] : 0.f;
    }
    Wp[t] = make_float4(v[0], v[1], v[2], v[3]);
}

// the chain of one quarter from packed weights already in registers: wv[j4] = the lane's weights of k-pairs 4 j4 .. 4 j4 + 3; np pairs
__device__ __forceinline__ void ds_chain_packed(const float4 (&wv)[DS_P4MAX], int np, const float* __restrict__ xs_row, int hh, ds_f32x16& acc) {
#pragma unroll
    for (int s7 = 0; s7 < DS_P4MAX; s7 += 7) {               // sets of 28 k-pairs, as ds_chain_chunk: the set's inputs out of LDS first, then its MFMAs
        const int sb = 4 * s7;
        if (sb < np) {                                       // uniform
            float xb[28];
#pragma unroll
            for (int j = 0; j < 28; ++j) xb[j] = xs_row[2 * min(sb + j, np - 1) + hh];
            __builtin_amdgcn_sched_barrier(0);
            if (sb + 28 <= np) {
#pragma unroll
                for (int j = 0; j < 28; ++j) {
                    if (s7 + j / 4 < DS_P4MAX) {
                        const float4& w4 = wv[(s7 + j / 4) < DS_P4MAX ? (s7 + j / 4) : 0];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j % 4 == 0 ? w4.x : j % 4 == 1 ? w4.y : j % 4 == 2 ? w4.z : w4.w, xb[j], acc, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 28; ++j) {
                    if (s7 + j / 4 < DS_P4MAX && sb + j < np) {
                        const float4& w4 = wv[(s7 + j / 4) < DS_P4MAX ? (s7 + j / 4) : 0];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(j % 4 == 0 ? w4.x : j % 4 == 1 ? w4.y : j % 4 == 2 ? w4.z : w4.w, xb[j], acc, 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifdef DS_TRACE         // development only (profiles/tools/det_step_trace.py): wall-clock stamps (100 MHz) of every wave of workgroup (0, 0, 0), [stage][wave]
__device__ long long ds_trace[8][4];
extern "C" int mnn_ds_trace_read(long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(ds_trace), sizeof(ds_trace)) == hipSuccess ? 0 : 1; }
#define DS_TR(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) ds_trace[k][threadIdx.x >> 6] = wall_clock64(); } while (0)
#else
#define DS_TR(k) do { } while (0)
#endif
__global__ void __launch_bounds__(256) lstm_step_det_kernel(DetLstmJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    DS_TR(0);
    const mnn_det_lstm_job& jb = J.job[blockIdx.z];
    const int u = jb.units, ub = blockIdx.x * 8;
    if (ub >= u) return;                             // the grid covers the widest job
    const int r0 = blockIdx.y * 32;
    const int n1 = jb.n_x, n2 = jb.n_x2;
    const int K = n1 + n2 + u;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;    // (w in a scalar register: the chain's bounds tests are then scalar branches, not an exec-mask dance around every MFMA)
    // every wave of the workgroup: the same 32 A rows = the four gates of units ub .. ub + 7 (row g * 8 + uu <-> TF column g * u + ub + uu);
    // wave w walks quarter w of K
    const int col = (r >> 3) * u + ub + (r & 7);
    const float* __restrict__ wp = jb.W + col;
    const size_t ldw = (size_t)4 * u;
    auto w_k = [&](int k) { return wp[(size_t)k * ldw]; };
    float* part = ds_smem;                           // [4 waves][16 registers][64 lanes]: the quarters' partial sums
    float* xs = ds_smem + DS_SPLIT * 16 * 64;
    ds_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float4* __restrict__ wpk = reinterpret_cast<const float4*>(jb.Wp);
    const int q4max = ds_pack_q4max(K), nch = ds_pack_nchunks(K);
    for (int k0 = 0; k0 < K; k0 += DS_KC) {
        const int kc = (min(DS_KC, K - k0) + 1) & ~1, pitch = kc | 1;
        float4 wv[DS_P4MAX];
        if (wpk != nullptr) {                            // this quarter's weights: every load in flight while the inputs are staged
            const float4* __restrict__ pq = wpk + ((size_t)((blockIdx.x * nch + k0 / DS_KC) * DS_SPLIT + w) * q4max) * 64 + lane;
#pragma unroll
            for (int j4 = 0; j4 < DS_P4MAX; ++j4) wv[j4] = pq[(size_t)min(j4, q4max - 1) * 64];
        }
        if (k0 > 0) __syncthreads();
        if (n1 > 0) {
            if (jb.x_dtype == MNN_U8) ds_stage_seg(reinterpret_cast<const uint8_t*>(jb.x), (size_t)jb.ld_x, (size_t)jb.es_x, n1, 0, B, r0, k0, kc, pitch, xs);
            else ds_stage_seg(reinterpret_cast<const float*>(jb.x), (size_t)jb.ld_x, (size_t)jb.es_x, n1, 0, B, r0, k0, kc, pitch, xs);
        }
        if (n2 > 0) ds_stage_seg(jb.x2, (size_t)jb.ld_x2, (size_t)1, n2, n1, B, r0, k0, kc, pitch, xs);
        if (jb.h_prev != nullptr) ds_stage_seg(jb.h_prev, (size_t)u, (size_t)1, u, n1 + n2, B, r0, k0, kc, pitch, xs);
        else ds_stage_zero(max(0, n1 + n2 - k0), min(kc, K - k0), pitch, xs);
        if (K - k0 < kc) ds_stage_zero(K - k0, kc, pitch, xs);               // the pad column of an odd K
        DS_TR(1);
        __syncthreads();
        DS_TR(2);
        const int q = ds_quarter(kc), p0 = min(w * q, kc / 2), p1 = min(p0 + q, kc / 2);
        if (p1 > p0) {
            if (wpk != nullptr) ds_chain_packed(wv, p1 - p0, xs + r * pitch + 2 * p0, hh, acc);
            else ds_chain_chunk(w_k, k0 + 2 * p0, 2 * (p1 - p0), K, xs + r * pitch + 2 * p0, hh, acc);
        }
        DS_TR(3);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[(w * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    DS_TR(4);
    // pointwise: thread -> (unit ub + (t & 7), row r0 + (t >> 3)); accumulator register of A row i = g * 8 + uu: e = 4 g + (uu & 3), lane half uu >> 2
    const int uu = threadIdx.x & 7, rr = threadIdx.x >> 3, row = r0 + rr, un = ub + uu;
    if (row >= B) return;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int at = (4 * g + (uu & 3)) * 64 + rr + 32 * (uu >> 2);
        const float s01 = __fadd_rn(part[at], part[16 * 64 + at]), s23 = __fadd_rn(part[2 * 16 * 64 + at], part[3 * 16 * 64 + at]);
        z[g] = __fadd_rn(__fadd_rn(s01, s23), jb.bias[g * u + un]);
    }
    const float gi = det_sigmoid(z[0]), gc = det_tanh(z[1]), gf = det_sigmoid(z[2]), go = det_sigmoid(z[3]);
    const float cp = jb.c_prev != nullptr ? jb.c_prev[(size_t)row * u + un] : 0.f;
    const float c = __fadd_rn(__fmul_rn(gc, gi), __fmul_rn(cp, gf));
    jb.c_out[(size_t)row * u + un] = c;
    jb.h_out[(size_t)row * u + un] = __fmul_rn(det_tanh(c), go);
    DS_TR(5);
}

__global__ void __launch_bounds__(256) dense_det_kernel(DetDenseJobs J, int B) {
    extern __shared__ __attribute__((aligned(16))) float ds_smem[];
    const mnn_det_dense_job& jb = J.job[blockIdx.z];
    const int nb = blockIdx.x * 32;
    if (nb >= jb.N) return;
    const int r0 = blockIdx.y * 32;
    const int K = jb.K;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, hh = lane >> 5;    // (w in a scalar register: the chain's bounds tests are then scalar branches, not an exec-mask dance around every MFMA)
    const int n = min(nb + r, jb.N - 1);             // this lane's A row = output column n (the same for the four waves: wave w walks quarter w of K)
    const float* __restrict__ wp = jb.W + n;
    const size_t ldw = (size_t)jb.ld_w;
    auto w_k = [&](int k) { return wp[(size_t)k * ldw]; };
    float* part = ds_smem;
    float* xs = ds_smem + DS_SPLIT * 16 * 64;
    ds_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float4* __restrict__ wpk = reinterpret_cast<const float4*>(jb.Wp);
    const int q4max = ds_pack_q4max(K), nch = ds_pack_nchunks(K);
    for (int k0 = 0; k0 < K; k0 += DS_KC) {
        const int kc = (min(DS_KC, K - k0) + 1) & ~1, pitch = kc | 1;
        float4 wv[DS_P4MAX];
        if (wpk != nullptr) {                            // packed weights (mnn_det_dense_pack): the quarter's loads all in flight, as in the LSTM step
            const float4* __restrict__ pq = wpk + ((size_t)((blockIdx.x * nch + k0 / DS_KC) * DS_SPLIT + w) * q4max) * 64 + lane;
#pragma unroll
            for (int j4 = 0; j4 < DS_P4MAX; ++j4) wv[j4] = pq[(size_t)min(j4, q4max - 1) * 64];
        }
        if (k0 > 0) __syncthreads();
        ds_stage_seg(jb.x, (size_t)jb.ld_x, (size_t)1, K, 0, B, r0, k0, kc, pitch, xs);
        if (K - k0 < kc) ds_stage_zero(K - k0, kc, pitch, xs);
        __syncthreads();
        const int q = ds_quarter(kc), p0 = min(w * q, kc / 2), p1 = min(p0 + q, kc / 2);
        if (p1 > p0) {
            if (wpk != nullptr) ds_chain_packed(wv, p1 - p0, xs + r * pitch + 2 * p0, hh, acc);
            else ds_chain_chunk(w_k, k0 + 2 * p0, 2 * (p1 - p0), K, xs + r * pitch + 2 * p0, hh, acc);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[(w * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    // thread -> (column nb + i, row r0 + (t >> 3)), i = (t & 7) + 8 pass: consecutive threads store consecutive columns
    const int rr = threadIdx.x >> 3, row = r0 + rr;
    if (row >= B) return;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int i = (threadIdx.x & 7) + 8 * pass, nn = nb + i;
        if (nn >= jb.N) continue;
        const int at = ((i & 3) + 4 * (i >> 3)) * 64 + rr + 32 * ((i >> 2) & 1);      // A row i = (e & 3) + 8 (e >> 2) + 4 hh
        const float s01 = __fadd_rn(part[at], part[16 * 64 + at]), s23 = __fadd_rn(part[2 * 16 * 64 + at], part[3 * 16 * 64 + at]);
        const float zz = __fadd_rn(s01, s23);
        jb.out[(size_t)row * jb.ld_out + nn] = jb.bias != nullptr ? __fadd_rn(zz, jb.bias[nn]) : zz;
    }
}

static size_t ds_lds_bytes(int K) { const int kc = (min(DS_KC, K) + 1) & ~1; return ((size_t)32 * (kc | 1) + DS_SPLIT * 16 * 64) * sizeof(float); }
static hipError_t ds_raise_lds() {
    static bool raised_[64];
    bool& raised = mnn_dev_flag(raised_);
    if (raised) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_step_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_det_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    raised = e == hipSuccess;
    return e;
}

extern "C" size_t mnn_det_lstm_pack_bytes(int K, int units) {
    if (K <= 0 || units <= 0 || units % 32 != 0) return 0;
    return (size_t)(units / 8) * ds_pack_nchunks(K) * DS_SPLIT * ds_pack_q4max(K) * 64 * sizeof(float4);
}
extern "C" int mnn_det_lstm_pack(mnn_stream_t s, const float* W, int K, int units, float* Wp) {
    MNN_REQUIRE(W && Wp && K > units && units > 0 && units % 32 == 0 && ((uintptr_t)Wp & 15) == 0,
                "mnn_det_lstm_pack: W [K, 4 units] with K = inputs + units, units %% 32 == 0, Wp 16-byte aligned (K=%d units=%d)", K, units);
    const long total = (long)(mnn_det_lstm_pack_bytes(K, units) / sizeof(float4));
    hipLaunchKernelGGL(ds_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, W, K, units, reinterpret_cast<float4*>(Wp), 0, 0);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" size_t mnn_det_dense_pack_bytes(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (size_t)((N + 31) / 32) * ds_pack_nchunks(K) * DS_SPLIT * ds_pack_q4max(K) * 64 * sizeof(float4);
}
extern "C" int mnn_det_dense_pack(mnn_stream_t s, const float* W, int K, int N, int ld_w, float* Wp) {
    MNN_REQUIRE(W && Wp && K > 0 && N > 0 && ld_w >= N && ((uintptr_t)Wp & 15) == 0, "mnn_det_dense_pack: W [K, ld_w >= N], Wp 16-byte aligned (K=%d N=%d)", K, N);
    const long total = (long)(mnn_det_dense_pack_bytes(K, N) / sizeof(float4));
    hipLaunchKernelGGL(ds_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, W, K, 0, reinterpret_cast<float4*>(Wp), N, ld_w);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
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
    dim3 grid(umax / 8, cdiv(B, 32), njobs);
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
    dim3 grid(cdiv(nmax, 32), cdiv(B, 32), njobs);
    hipLaunchKernelGGL(dense_det_kernel, grid, dim3(256), ds_lds_bytes(kmax), (hipStream_t)s, J, B);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The whole sampling scan of an LSTM-NADE generator as ONE call (rnn_estimator.py:271-323 `generate` / `_generate_recurrence`;
// rnn_nade.py:253-277, 304-318; SURVEY.md 8(b) mnn_generate_scan): intro pass -- n_intro deterministic steps of the LSTM stack from a zero
// state, Dense on the last output -- then num_steps x { NADE sample (mnn_nade_sample, Philox sub-counter = the generated step) -> LSTM step on
// the sample -> Dense }.  Everything is enqueued on the caller's stream from this one host loop (4 launches per generated step for a
// two-layer stack; nothing is synchronised), so the call is capturable into a hipGraph like any other entry point; states ping-pong in the
// caller's workspace.  Bits: exactly those of the single-step entry points above (oracle/det.py rnn_nade_generate restates the scan).
// ------------------------------------------------------------------------------------------------------------------------------------
static size_t scan_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t mnn_generate_scan_workspace_bytes(int B, int n_in, int n_layers, const mnn_scan_lstm_layer* layers, int n_out) {
    if (B <= 0 || n_in <= 0 || n_layers <= 0 || n_layers > MNN_SCAN_MAX_LAYERS || layers == nullptr || n_out <= 0) return 0;
    size_t bytes = scan_align((size_t)B * (size_t)((n_out + 63) & ~63) * sizeof(float));
    for (int l = 0; l < n_layers; ++l) {
        bytes += 4 * scan_align((size_t)B * (size_t)layers[l].units * sizeof(float));      // c, h x two generations
        bytes += scan_align(mnn_det_lstm_pack_bytes((l == 0 ? n_in : layers[l - 1].units) + layers[l].units, layers[l].units));      // the layer's packed weights
    }
    bytes += scan_align(mnn_det_dense_pack_bytes(layers[n_layers - 1].units, n_out));      // the Dense kernel's
    return bytes;
}

extern "C" int mnn_generate_scan(mnn_stream_t s, int B, int n_intro, int num_steps, const uint8_t* intro, int n_in, int n_layers,
                                 const mnn_scan_lstm_layer* layers, const float* dense_W, const float* dense_bias, int n_out, int tracks, int D,
                                 int Hn, const float* w_enc, const float* w_dec, float temperature, uint64_t seed, uint32_t row0,
                                 uint8_t* samples, void* workspace, size_t workspace_bytes) {
    MNN_REQUIRE(B > 0 && n_intro > 0 && num_steps >= 0 && intro && n_in > 0 && n_layers > 0 && n_layers <= MNN_SCAN_MAX_LAYERS && layers,
                "mnn_generate_scan: B, n_intro > 0, 1..%d layers", MNN_SCAN_MAX_LAYERS);
    MNN_REQUIRE(dense_W && tracks > 0 && D > 0 && Hn > 0 && n_out == tracks * (Hn + D) && n_in == tracks * D && w_enc && w_dec && samples,
                "mnn_generate_scan: the Dense layer feeds `tracks` NADEs (n_out == tracks * (Hn + D)) and a sample is the next input (n_in == tracks * D)");
    const size_t need = mnn_generate_scan_workspace_bytes(B, n_in, n_layers, layers, n_out);
    MNN_REQUIRE(workspace && need > 0 && workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0,
                "mnn_generate_scan: workspace of mnn_generate_scan_workspace_bytes() bytes, 256-byte aligned");
    char* wp = static_cast<char*>(workspace);
    const int ld_out = (n_out + 63) & ~63;
    float* out = reinterpret_cast<float*>(wp);
    wp += scan_align((size_t)B * ld_out * sizeof(float));
    float* cbuf[MNN_SCAN_MAX_LAYERS][2];
    float* hbuf[MNN_SCAN_MAX_LAYERS][2];
    for (int l = 0; l < n_layers; ++l) {
        MNN_REQUIRE(layers[l].units > 0 && layers[l].units % 32 == 0 && layers[l].W && layers[l].bias, "mnn_generate_scan: layer %d", l);
        for (int g = 0; g < 2; ++g) {
            cbuf[l][g] = reinterpret_cast<float*>(wp); wp += scan_align((size_t)B * layers[l].units * sizeof(float));
            hbuf[l][g] = reinterpret_cast<float*>(wp); wp += scan_align((size_t)B * layers[l].units * sizeof(float));
        }
    }
    // the master weights repacked once per scan for the step kernel's 16-byte loads (mnn_det_lstm_pack: same numbers, same chains)
    float* wpack[MNN_SCAN_MAX_LAYERS];
    for (int l = 0; l < n_layers; ++l) {
        const int K = (l == 0 ? n_in : layers[l - 1].units) + layers[l].units;
        wpack[l] = reinterpret_cast<float*>(wp);
        wp += scan_align(mnn_det_lstm_pack_bytes(K, layers[l].units));
        const int rc = mnn_det_lstm_pack(s, layers[l].W, K, layers[l].units, wpack[l]);
        if (rc != MNN_OK) return rc;
    }
    float* dpack = reinterpret_cast<float*>(wp);
    wp += scan_align(mnn_det_dense_pack_bytes(layers[n_layers - 1].units, n_out));
    {
        const int rc = mnn_det_dense_pack(s, dense_W, layers[n_layers - 1].units, n_out, n_out, dpack);
        if (rc != MNN_OK) return rc;
    }
    int cur = 0;                                               // generation holding the current state; -1 before the first step (zero state)
    bool have_state = false;
    auto stack_step = [&](const uint8_t* x, int ld_x) -> int {      // one step of the whole stack on a u8 input block
        const int nxt = cur ^ 1;
        for (int l = 0; l < n_layers; ++l) {
            mnn_det_lstm_job jb;
            memset(&jb, 0, sizeof(jb));
            const int u = layers[l].units;
            if (l == 0) { jb.x = x; jb.x_dtype = MNN_U8; jb.n_x = n_in; jb.ld_x = ld_x; jb.es_x = 1; }
            else { jb.x = hbuf[l - 1][nxt]; jb.x_dtype = MNN_F32; jb.n_x = layers[l - 1].units; jb.ld_x = layers[l - 1].units; jb.es_x = 1; }
            jb.h_prev = have_state ? hbuf[l][cur] : nullptr;
            jb.c_prev = have_state ? cbuf[l][cur] : nullptr;
            jb.W = layers[l].W; jb.Wp = wpack[l]; jb.bias = layers[l].bias; jb.c_out = cbuf[l][nxt]; jb.h_out = hbuf[l][nxt]; jb.units = u;
            const int rc = mnn_lstm_step_det(s, B, 1, &jb);
            if (rc != MNN_OK) return rc;
        }
        cur = nxt;
        have_state = true;
        return MNN_OK;
    };
    auto dense = [&]() -> int {
        mnn_det_dense_job dj;
        memset(&dj, 0, sizeof(dj));
        const int ul = layers[n_layers - 1].units;
        dj.x = hbuf[n_layers - 1][cur]; dj.ld_x = ul; dj.K = ul; dj.W = dense_W; dj.Wp = dpack; dj.ld_w = n_out; dj.N = n_out; dj.bias = dense_bias;
        dj.out = out; dj.ld_out = ld_out;
        return mnn_dense_det(s, B, 1, &dj);
    };
    for (int t = 0; t < n_intro; ++t) {
        const int rc = stack_step(intro + (size_t)t * n_in, n_intro * n_in);
        if (rc != MNN_OK) return rc;
    }
    int rc = dense();
    if (rc != MNN_OK) return rc;
    const int row_stride = num_steps * n_in;
    for (int st = 0; st < num_steps; ++st) {
        uint8_t* smp = samples + (size_t)st * n_in;            // samples[:, st, :]; feature m D + i (one NADE) or i tracks + m (rnn_multinade.py:313-314)
        rc = mnn_nade_sample(s, tracks, B, D, Hn, out, ld_out, w_enc, w_dec, temperature, seed, row0, (uint32_t)st, smp,
                             tracks > 1 ? 1 : D, row_stride, tracks > 1 ? tracks : 1, nullptr);
        if (rc != MNN_OK) return rc;
        rc = stack_step(smp, row_stride);
        if (rc != MNN_OK) return rc;
        rc = dense();
        if (rc != MNN_OK) return rc;
    }
    return MNN_OK;
}
