// HBM-bound plumbing kernels: layout changes, dropout, reductions, optimiser.
#include <stdarg.h>

#include "common.h"

// ----------------------------------------------------------------------------------------------
// housekeeping
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void mnn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mnn_last_error(void) { return g_err; }
extern "C" int mnn_version(void) { return MNN_ABI_VERSION; }

template <typename T> __device__ __forceinline__ float ld_as_f32(const T* p, size_t i);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p, size_t i) { return bf16_to_f32(p[i]); }
template <> __device__ __forceinline__ float ld_as_f32<f16_t>(const f16_t* p, size_t i) { return (float)p[i]; }
template <> __device__ __forceinline__ float ld_as_f32<uint8_t>(const uint8_t* p, size_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st_from_f32(T* p, size_t i, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(float* p, size_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, size_t i, float v) { p[i] = f32_to_bf16(v); }
template <> __device__ __forceinline__ void st_from_f32<f16_t>(f16_t* p, size_t i, float v) { p[i] = (f16_t)v; }

// ----------------------------------------------------------------------------------------------
// transpose with conversion: out[c, r] = in[r, c]; 32x32 tiles through LDS (+1 pad)
// ----------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) transpose_kernel(const TI* __restrict__ in, int R, int C, int ld_in, TO* __restrict__ out, int ld_out) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, c = c0 + tx;
        tile[ty + k * 8][tx] = (r < R && c < C) ? ld_as_f32<TI>(in, (size_t)r * ld_in + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + k * 8, r = r0 + tx;
        if (c < C && r < R) st_from_f32<TO>(out, (size_t)c * ld_out + r, tile[tx][ty + k * 8]);
    }
}

template <typename TI, typename TO>
static int launch_transpose(hipStream_t st, const void* in, int R, int C, int ld_in, void* out, int ld_out) {
    dim3 grid(cdiv(C, 32), cdiv(R, 32));
    hipLaunchKernelGGL((transpose_kernel<TI, TO>), grid, dim3(256), 0, st, (const TI*)in, R, C, ld_in, (TO*)out, ld_out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_transpose(mnn_stream_t s, const void* in, int in_dtype, int R, int C, int ld_in, void* out, int out_dtype, int ld_out) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(in && out && R > 0 && C > 0 && ld_in >= C && ld_out >= R, "mnn_transpose: bad arguments R=%d C=%d ld_in=%d ld_out=%d", R, C,
                ld_in, ld_out);
    MNN_REQUIRE(out_dtype == MNN_F32 || out_dtype == MNN_BF16 || out_dtype == MNN_F16, "mnn_transpose: out dtype must be f32/bf16/f16");
#define TR(TI) (out_dtype == MNN_F32 ? launch_transpose<TI, float>(st, in, R, C, ld_in, out, ld_out)          \
                : out_dtype == MNN_F16 ? launch_transpose<TI, f16_t>(st, in, R, C, ld_in, out, ld_out)         \
                                       : launch_transpose<TI, bf16_t>(st, in, R, C, ld_in, out, ld_out))
    if (in_dtype == MNN_F32) return TR(float);
    if (in_dtype == MNN_BF16) return TR(bf16_t);
    if (in_dtype == MNN_F16) return TR(f16_t);
    if (in_dtype == MNN_U8) return TR(uint8_t);
#undef TR
    mnn_set_error("mnn_transpose: unknown in dtype %d", in_dtype);
    return MNN_ERR_INVALID;
}

template <typename TI, typename TO>
__global__ void convert2d_kernel(const TI* __restrict__ src, int ld_src, TO* __restrict__ dst, int ld_dst, int R, int C) {
    const long n = (long)R * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        st_from_f32<TO>(dst, (size_t)r * ld_dst + c, ld_as_f32<TI>(src, (size_t)r * ld_src + c));
    }
}

extern "C" int mnn_convert2d(mnn_stream_t s, const void* src, int src_dtype, int ld_src, void* dst, int dst_dtype, int ld_dst, int R, int C) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(src && dst && R > 0 && C > 0 && ld_src >= C && ld_dst >= C, "mnn_convert2d: bad arguments");
    MNN_REQUIRE(dst_dtype == MNN_F32 || dst_dtype == MNN_BF16 || dst_dtype == MNN_F16, "mnn_convert2d: dst dtype must be f32/bf16/f16");
    const int blocks = (int)min((long)2048, ((long)R * C + 255) / 256);
#define CV(TI)                                                                                                              \
    do {                                                                                                                    \
        if (dst_dtype == MNN_F32)                                                                                           \
            hipLaunchKernelGGL((convert2d_kernel<TI, float>), dim3(blocks), dim3(256), 0, st, (const TI*)src, ld_src, (float*)dst, ld_dst, R, C); \
        else if (dst_dtype == MNN_F16)                                                                                      \
            hipLaunchKernelGGL((convert2d_kernel<TI, f16_t>), dim3(blocks), dim3(256), 0, st, (const TI*)src, ld_src, (f16_t*)dst, ld_dst, R, C); \
        else                                                                                                                \
            hipLaunchKernelGGL((convert2d_kernel<TI, bf16_t>), dim3(blocks), dim3(256), 0, st, (const TI*)src, ld_src, (bf16_t*)dst, ld_dst, R, C); \
    } while (0)
    if (src_dtype == MNN_F32) CV(float);
    else if (src_dtype == MNN_BF16) CV(bf16_t);
    else if (src_dtype == MNN_F16) CV(f16_t);
    else if (src_dtype == MNN_U8) CV(uint8_t);
    else { mnn_set_error("mnn_convert2d: unknown src dtype %d", src_dtype); return MNN_ERR_INVALID; }
#undef CV
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// piano-roll plumbing (multinn_joint.py:83-89,132-139)
// one block per (t, b) row; threads sweep the D features
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(128) pianoroll_shift_kernel(const uint8_t* __restrict__ x, int B, int Tn, int D,
                                                             const int32_t* __restrict__ lengths, T* __restrict__ inputs, int ld_in,
                                                             uint8_t* __restrict__ targets, float* __restrict__ row_weight,
                                                             float inv_n) {
    const int t = blockIdx.x / B, b = blockIdx.x % B;
    const uint8_t* cur = x + ((size_t)b * Tn + t) * D;
    const uint8_t* prev = t > 0 ? cur - D : nullptr;
    T* in = inputs + (size_t)blockIdx.x * ld_in;
    uint8_t* tg = targets ? targets + (size_t)blockIdx.x * D : nullptr;
    for (int i = threadIdx.x; i < ld_in; i += blockDim.x) {
        const float pv = (prev != nullptr && i < D) ? (float)prev[i] : 0.f;
        st_from_f32<T>(in, i, pv);
        if (tg != nullptr && i < D) tg[i] = cur[i];
    }
    if (threadIdx.x == 0 && row_weight != nullptr) row_weight[blockIdx.x] = (lengths == nullptr || t < lengths[b]) ? inv_n : 0.f;
}

extern "C" int mnn_pianoroll_shift_timemajor(mnn_stream_t s, const uint8_t* x, int B, int T, int D, const int32_t* lengths, void* inputs,
                                             int in_dtype, int ld_in, uint8_t* targets, float* row_weight, long n_valid_total) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(x && inputs && B > 0 && T > 0 && D > 0 && ld_in >= D, "mnn_pianoroll_shift_timemajor: bad arguments");
    MNN_REQUIRE(in_dtype == MNN_F32 || in_dtype == MNN_BF16 || in_dtype == MNN_F16, "mnn_pianoroll_shift_timemajor: inputs dtype must be f32/bf16/f16");
    MNN_REQUIRE(lengths == nullptr || n_valid_total > 0, "mnn_pianoroll_shift_timemajor: n_valid_total required with lengths");
    const float inv_n = 1.0f / (float)(n_valid_total > 0 ? n_valid_total : (long)B * T);
    if (in_dtype == MNN_F32)
        hipLaunchKernelGGL(pianoroll_shift_kernel<float>, dim3(B * T), dim3(128), 0, st, x, B, T, D, lengths, (float*)inputs, ld_in, targets,
                           row_weight, inv_n);
    else if (in_dtype == MNN_F16)
        hipLaunchKernelGGL(pianoroll_shift_kernel<f16_t>, dim3(B * T), dim3(128), 0, st, x, B, T, D, lengths, (f16_t*)inputs, ld_in, targets,
                           row_weight, inv_n);
    else
        hipLaunchKernelGGL(pianoroll_shift_kernel<bf16_t>, dim3(B * T), dim3(128), 0, st, x, B, T, D, lengths, (bf16_t*)inputs, ld_in, targets,
                           row_weight, inv_n);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// Tiled form for the bf16 train step: 64 (t, b) rows x 64 features per workgroup, 8-byte loads of the piano-roll, 16-byte
// stores of the shifted inputs, and ALSO the transposed copy inputs_t[feature][t B + b] (the K-major operand of layer 1's
// weight-gradient GEMM) through an LDS tile -- instead of a second pass (mnn_transpose) over the 29 MB of inputs.
template <typename F>
__global__ void __launch_bounds__(256)
pianoroll_shift_tiled_kernel(const uint8_t* __restrict__ x, int B, int Tn, int D, const int32_t* __restrict__ lengths,
                             bf16_t* __restrict__ inputs, int ld_in, bf16_t* __restrict__ inputs_t, int ld_t,
                             uint8_t* __restrict__ targets, float* __restrict__ row_weight, float inv_n, unsigned* __restrict__ count,
                             const int32_t* __restrict__ inv, const int32_t* __restrict__ hdr) {
    __shared__ bf16_t tile[64][66];
    unsigned nset = 0;                                   // set cells among the target bytes this thread writes (the NADE forward's density gate)
    const int N = B * Tn;
    const int n0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int fq = threadIdx.x & 7, f = f0 + 8 * fq;
    const bool vec = (D & 7) == 0;                       // 8-byte source loads / target stores stay aligned
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rr = (threadIdx.x >> 3) + 32 * k, n = n0 + rr;
        uint8_t pv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (n < N && f < D) {
            const int t = n / B, b = n - t * B;
            const uint8_t* cur = x + ((size_t)b * Tn + t) * D + f;
            if (vec) {                                   // f + 8 <= D because D and f are multiples of 8
                *reinterpret_cast<uint2*>(cv) = *reinterpret_cast<const uint2*>(cur);
                if (t > 0) *reinterpret_cast<uint2*>(pv) = *reinterpret_cast<const uint2*>(cur - D);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (f + e < D) { cv[e] = cur[e]; pv[e] = t > 0 ? cur[e - D] : (uint8_t)0; }
            }
        }
        bf16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = F::cvt((float)pv[e]);                    // 0..255 are exact in bf16 and in f16
            tile[rr][8 * fq + e] = o[e];
        }
        if (n < N) {
            if (f + 8 <= ld_in) {
                uint4 q;
                q.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16); q.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
                q.z = (uint32_t)o[4] | ((uint32_t)o[5] << 16); q.w = (uint32_t)o[6] | ((uint32_t)o[7] << 16);
                *reinterpret_cast<uint4*>(inputs + (size_t)n * ld_in + f) = q;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (f + e < ld_in) inputs[(size_t)n * ld_in + f + e] = o[e];
            }
            const int nc = inv != nullptr ? inv[n] : n;                     // compact row order (mnn_ragged_index): targets and weights only
            if (targets != nullptr && f < D) {
#pragma unroll
                for (int e = 0; e < 8; ++e) nset += cv[e] != 0;
                if (vec) *reinterpret_cast<uint2*>(targets + (size_t)nc * D + f) = *reinterpret_cast<uint2*>(cv);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (f + e < D) targets[(size_t)nc * D + f + e] = cv[e];
                }
            }
            if (row_weight != nullptr && blockIdx.x == 0 && fq == 0) {
                if (inv != nullptr) row_weight[nc] = nc < hdr[0] ? __int_as_float(hdr[1]) : 0.f;
                else {
                    const int t = n / B, b = n - t * B;
                    row_weight[n] = (lengths == nullptr || t < lengths[b]) ? inv_n : 0.f;
                }
            }
        }
    }
    if (count != nullptr) {                              // one atomic per wave: the count mnn_density_gate would take in a second pass over the targets
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nset += __shfl_xor(nset, o);
        // spread over MNN_DENSITY_SLOTS words: 114 k same-address atomics (one per wave at [1024,256,88,5]) serialise -- measured 1.3 ms for this pass
        if ((threadIdx.x & 63) == 0 && nset != 0u) atomicAdd(count + ((blockIdx.y * gridDim.x + blockIdx.x) & (MNN_DENSITY_SLOTS - 1)), nset);
    }
    __syncthreads();
    const int tc = threadIdx.x >> 2, rq = threadIdx.x & 3;
    if (f0 + tc < ld_in) {
        bf16_t* dst = inputs_t + (size_t)(f0 + tc) * ld_t + n0 + 16 * rq;
        if (n0 + 16 * rq + 15 < N) {
            uint32_t w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (uint32_t)tile[16 * rq + 2 * e][tc] | ((uint32_t)tile[16 * rq + 2 * e + 1][tc] << 16);
            reinterpret_cast<uint4*>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
            reinterpret_cast<uint4*>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (n0 + 16 * rq + e < N) dst[e] = tile[16 * rq + e][tc];
        }
    }
}

extern "C" int mnn_pianoroll_shift_timemajor_t(mnn_stream_t s, const uint8_t* x, int B, int T, int D, const int32_t* lengths, void* inputs,
                                               int ld_in, void* inputs_t, int ld_t, uint8_t* targets, float* row_weight, long n_valid_total, int dtype,
                                               unsigned* count, const int32_t* inv, const int32_t* hdr) {
    MNN_REQUIRE((inv == nullptr) == (hdr == nullptr), "mnn_pianoroll_shift_timemajor_t: inv and hdr (mnn_ragged_index) come together");
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16, "mnn_pianoroll_shift_timemajor_t: dtype must be bf16 or f16");
    MNN_REQUIRE(x && inputs && inputs_t && B > 0 && T > 0 && D > 0 && ld_in >= D, "mnn_pianoroll_shift_timemajor_t: bad arguments");
    MNN_REQUIRE(ld_in % 8 == 0 && ld_t % 8 == 0 && ld_t >= B * T, "mnn_pianoroll_shift_timemajor_t: ld_in, ld_t must be multiples of 8, ld_t >= B*T");
    MNN_REQUIRE(((uintptr_t)inputs & 15) == 0 && ((uintptr_t)inputs_t & 15) == 0 && ((uintptr_t)x & 7) == 0 && (targets == nullptr || ((uintptr_t)targets & 7) == 0),
                "mnn_pianoroll_shift_timemajor_t: misaligned buffer");
    MNN_REQUIRE(lengths == nullptr || n_valid_total > 0 || inv != nullptr, "mnn_pianoroll_shift_timemajor_t: n_valid_total required with lengths");
    const float inv_n = 1.0f / (float)(n_valid_total > 0 ? n_valid_total : (long)B * T);
    dim3 grid(cdiv(ld_in, 64), cdiv((long)B * T, 64));
    if (dtype == MNN_F16)
        hipLaunchKernelGGL(pianoroll_shift_tiled_kernel<Fp16F>, grid, dim3(256), 0, (hipStream_t)s, x, B, T, D, lengths, (bf16_t*)inputs, ld_in,
                           (bf16_t*)inputs_t, ld_t, targets, row_weight, inv_n, count, inv, hdr);
    else
        hipLaunchKernelGGL(pianoroll_shift_tiled_kernel<Bf16F>, grid, dim3(256), 0, (hipStream_t)s, x, B, T, D, lengths, (bf16_t*)inputs, ld_in,
                           (bf16_t*)inputs_t, ld_t, targets, row_weight, inv_n, count, inv, hdr);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// Ragged windows: compaction of the valid rows (utils/sequences.py:6-37; see include/multinn_hip.h "Ragged windows")
// ----------------------------------------------------------------------------------------------
// One workgroup per timestep t.  Valid rows before slab t: sum_b min(len_b, t); inside the slab: a ballot prefix over b.
__global__ void __launch_bounds__(256) ragged_index_kernel(const int32_t* __restrict__ lengths, int B, int T, const float* __restrict__ n_total_dev,
                                                           float scale_rows, int32_t* __restrict__ idx, int32_t* __restrict__ inv, int32_t* __restrict__ hdr) {
    __shared__ int s_red[2][4];
    __shared__ int s_wave[4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int before = 0, total = 0;
    for (int b = tid; b < B; b += 256) {
        const int len = min(max(lengths[b], 0), T);
        before += min(len, t);
        total += len;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); total += __shfl_xor(total, o); }
    if (lane == 0) { s_red[0][w] = before; s_red[1][w] = total; }
    __syncthreads();
    before = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
    total = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    if (t == 0 && tid == 0) {
        const float ntot = n_total_dev != nullptr ? fmaxf(*n_total_dev, 1.f) : (float)max(total, 1);
        const float ls = scale_rows > 0.f ? exp2f(rintf(log2f(scale_rows * ntot))) : 1.f;
        hdr[0] = total;
        hdr[1] = __float_as_int(1.0f / ntot);
        hdr[2] = __float_as_int(ls);
        hdr[3] = __float_as_int(1.0f / ls);
    }
    int run = 0;                                             // valid rows of this slab in front of the current chunk of 256
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + tid;
        const bool ok = b < B && t < lengths[min(b, B - 1)];
        const unsigned long long bal = __ballot(ok);
        __syncthreads();                                     // s_wave of the previous chunk has been read
        if (lane == 0) s_wave[w] = __popcll(bal);
        __syncthreads();
        int pre = 0;
        for (int ww = 0; ww < w; ++ww) pre += s_wave[ww];
        const int rank = run + pre + __popcll(bal & ((1ull << lane) - 1ull));
        if (b < B) {
            const int row = t * B + b;
            const int k = ok ? before + rank : total + (row - (before + rank));
            idx[k] = row;
            inv[row] = k;
        }
        run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    }
}
extern "C" int mnn_ragged_index(mnn_stream_t s, const int32_t* lengths, int B, int T, const float* n_total_dev, float scale_rows, int32_t* idx, int32_t* inv,
                                int32_t* hdr) {
    MNN_REQUIRE(lengths && idx && inv && hdr && B > 0 && T > 0 && (long)B * T < (1L << 31), "mnn_ragged_index: bad arguments");
    hipLaunchKernelGGL(ragged_index_kernel, dim3(T), dim3(256), 0, (hipStream_t)s, lengths, B, T, n_total_dev, scale_rows, idx, inv, hdr);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// dst[k] = k < n_rows ? src[idx[k]] : 0 for 16-bit rows of C elements (C % 8 == 0), 64 x 64 tiles; optionally also the transposed copy
// dst_t[c][k] (through an LDS tile: 32-byte runs per column).
__global__ void __launch_bounds__(256) rows_gather16_kernel(const uint16_t* __restrict__ src, int ld_src, const int32_t* __restrict__ idx,
                                                            const int32_t* __restrict__ n_rows_dev, int N, int C, uint16_t* __restrict__ dst, int ld_dst,
                                                            uint16_t* __restrict__ dst_t, int ld_t) {
    __shared__ uint16_t tile[64][66];
    const int n0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int nv = n_rows_dev != nullptr ? min(*n_rows_dev, N) : N;
    const int fq = threadIdx.x & 7, f = f0 + 8 * fq;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rr = (threadIdx.x >> 3) + 32 * k, n = n0 + rr;
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if (n < nv && f < C) q = *reinterpret_cast<const uint4*>(src + (size_t)idx[n] * ld_src + f);
        if (n < N && f < C) *reinterpret_cast<uint4*>(dst + (size_t)n * ld_dst + f) = q;
        const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[rr][8 * fq + 2 * e] = (uint16_t)(wq[e] & 0xffffu);
            tile[rr][8 * fq + 2 * e + 1] = (uint16_t)(wq[e] >> 16);
        }
    }
    if (dst_t == nullptr) return;
    __syncthreads();
    const int tc = threadIdx.x >> 2, rq = threadIdx.x & 3;
    if (f0 + tc < C) {
        uint16_t* o = dst_t + (size_t)(f0 + tc) * ld_t + n0 + 16 * rq;
        if (n0 + 16 * rq + 15 < N) {
            uint32_t wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = (uint32_t)tile[16 * rq + 2 * e][tc] | ((uint32_t)tile[16 * rq + 2 * e + 1][tc] << 16);
            reinterpret_cast<uint4*>(o)[0] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
            reinterpret_cast<uint4*>(o)[1] = make_uint4(wv[4], wv[5], wv[6], wv[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (n0 + 16 * rq + e < N) o[e] = tile[16 * rq + e][tc];
        }
    }
}
extern "C" int mnn_rows_gather16(mnn_stream_t s, const void* src, int ld_src, const int32_t* idx, const int32_t* n_rows_dev, int N, int C, void* dst,
                                 int ld_dst, void* dst_t, int ld_t) {
    MNN_REQUIRE(src && idx && dst && N > 0 && C > 0 && C % 8 == 0 && ld_src >= C && ld_dst >= C && ld_src % 8 == 0 && ld_dst % 8 == 0,
                "mnn_rows_gather16: C and the pitches must be multiples of 8");
    MNN_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0 && (dst_t == nullptr || (((uintptr_t)dst_t & 15) == 0 && ld_t >= N && ld_t % 8 == 0)),
                "mnn_rows_gather16: misaligned buffer / ld_t");
    hipLaunchKernelGGL(rows_gather16_kernel, dim3(cdiv(C, 64), cdiv(N, 64)), dim3(256), 0, (hipStream_t)s, (const uint16_t*)src, ld_src, idx, n_rows_dev, N, C,
                       (uint16_t*)dst, ld_dst, (uint16_t*)dst_t, ld_t);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// dst[row] = inv[row] < n_rows ? src[inv[row]] : 0 (f32 rows of C elements, C % 4 == 0): the Dense input gradient back in time-major order
__global__ void __launch_bounds__(256) rows_scatter_f32_kernel(const float4* __restrict__ src, const int32_t* __restrict__ inv, const int32_t* __restrict__ n_rows_dev,
                                                               int N, int C4, float4* __restrict__ dst) {
    const int nv = n_rows_dev != nullptr ? min(*n_rows_dev, N) : N;
    const long total = (long)N * C4;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int row = (int)(e / C4), c = (int)(e - (long)row * C4);
        const int k = inv[row];
        dst[e] = k < nv ? src[(size_t)k * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
extern "C" int mnn_rows_scatter_f32(mnn_stream_t s, const float* src, const int32_t* inv, const int32_t* n_rows_dev, int N, int C, float* dst) {
    MNN_REQUIRE(src && inv && dst && N > 0 && C > 0 && C % 4 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0,
                "mnn_rows_scatter_f32: C must be a multiple of 4, buffers 16-byte aligned");
    const long total = (long)N * (C / 4);
    hipLaunchKernelGGL(rows_scatter_f32_kernel, dim3((int)std::min(16384L, (total + 255) / 256)), dim3(256), 0, (hipStream_t)s, (const float4*)src, inv, n_rows_dev,
                       N, C / 4, (float4*)dst);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

__global__ void __launch_bounds__(128) split_tracks_kernel(const uint8_t* __restrict__ x, int B, int Tn, int P, int M, uint8_t* __restrict__ out) {
    const int t = blockIdx.x / B, b = blockIdx.x % B;
    const uint8_t* src = x + ((size_t)b * Tn + t) * P * M;
    for (int i = threadIdx.x; i < P * M; i += blockDim.x) {
        const int p = i / M, m = i % M;
        out[((size_t)m * Tn * B + blockIdx.x) * P + p] = src[i];
    }
}

extern "C" int mnn_pianoroll_split_tracks(mnn_stream_t s, const uint8_t* x, int B, int T, int P, int M, uint8_t* targets_tracks) {
    MNN_REQUIRE(x && targets_tracks && B > 0 && T > 0 && P > 0 && M > 0, "mnn_pianoroll_split_tracks: bad arguments");
    hipLaunchKernelGGL(split_tracks_kernel, dim3(B * T), dim3(128), 0, (hipStream_t)s, x, B, T, P, M, targets_tracks);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// LSTM weight packing: natural TF kernel W[(in+u), 4u] -> gate-interleaved K-contiguous copies
// ----------------------------------------------------------------------------------------------
// 32 packed columns (one gate of one 32-unit block = 32 consecutive natural columns) x 32 rows of W per workgroup; W is read along
// the natural columns, wx_p / wh_p written along the packed columns and wx_t / wh_t (transposed) along k through an LDS tile: every
// access coalesced (an element-per-thread form scattered the transposed copies with a stride of ld_in elements).
template <typename T>
__global__ void __launch_bounds__(256)
lstm_pack_tiled_kernel(const float* __restrict__ W, const float* __restrict__ bias, int n_in, int U, int ld_in, T* __restrict__ wx_t,
                       T* __restrict__ wh_t, T* __restrict__ wh_p, T* __restrict__ wx_p, float* __restrict__ bias_p) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int pc0 = blockIdx.x * 32, ub = blockIdx.x >> 2, g = blockIdx.x & 3;
    const int nat0 = g * U + ub * 32, N4 = 4 * U, k0 = blockIdx.y * 32;
    const int kmax = n_in + U;              // rows of W; k in [n_in, ld_in) of wx_t is zero padding (rows k0.. may cover it)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int k = k0 + ty + 8 * jj;
        const float wv = k < kmax ? W[(size_t)k * N4 + nat0 + tx] : 0.f;
        tile[ty + 8 * jj][tx] = wv;
        if (k < n_in) { if (wx_p != nullptr) st_from_f32<T>(wx_p, (size_t)k * N4 + pc0 + tx, wv); }
        else if (k < kmax) st_from_f32<T>(wh_p, (size_t)(k - n_in) * N4 + pc0 + tx, wv);
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int pc = pc0 + ty + 8 * jj, k = k0 + tx;
        const float wv = tile[tx][ty + 8 * jj];
        if (k < n_in) st_from_f32<T>(wx_t, (size_t)pc * ld_in + k, wv);
        else if (k < kmax) st_from_f32<T>(wh_t, (size_t)pc * U + (k - n_in), wv);
    }
    if (blockIdx.y == 0) {
        if (ty == 0) bias_p[pc0 + tx] = bias[nat0 + tx];
        // zero the K padding of this workgroup's 32 rows of wx_t
        const int pad = ld_in - n_in;
        for (int i = threadIdx.x; i < 32 * pad; i += 256) st_from_f32<T>(wx_t, (size_t)(pc0 + i / pad) * ld_in + n_in + i % pad, 0.f);
    }
}

extern "C" int mnn_lstm_pack_weights(mnn_stream_t s, const float* W, const float* bias, int n_in, int units, int dtype, int ld_in,
                                     void* wx_t, void* wh_t, void* wh_p, void* wx_p, float* bias_p) {
    MNN_REQUIRE(W && bias && wx_t && wh_t && wh_p && bias_p, "mnn_lstm_pack_weights: null pointer");
    MNN_REQUIRE(units > 0 && units % 32 == 0 && n_in > 0 && ld_in >= n_in, "mnn_lstm_pack_weights: units %% 32 != 0 or bad n_in/ld_in");
    MNN_REQUIRE(dtype == MNN_F32 || dtype == MNN_BF16 || dtype == MNN_F16, "mnn_lstm_pack_weights: dtype must be f32/bf16/f16");
    dim3 grid(4 * units / 32, cdiv(n_in + units, 32));
    if (dtype == MNN_F32)
        hipLaunchKernelGGL(lstm_pack_tiled_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, W, bias, n_in, units, ld_in, (float*)wx_t,
                           (float*)wh_t, (float*)wh_p, (float*)wx_p, bias_p);
    else if (dtype == MNN_F16)
        hipLaunchKernelGGL(lstm_pack_tiled_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, W, bias, n_in, units, ld_in, (f16_t*)wx_t,
                           (f16_t*)wh_t, (f16_t*)wh_p, (f16_t*)wx_p, bias_p);
    else
        hipLaunchKernelGGL(lstm_pack_tiled_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, W, bias, n_in, units, ld_in, (bf16_t*)wx_t,
                           (bf16_t*)wh_t, (bf16_t*)wh_p, (bf16_t*)wx_p, bias_p);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// Rows of a gate-interleaved [4u, ld] matrix (row (unit/32)*128 + g*32 + unit%32) re-ordered GATE-MINOR (row unit*4 + g), and
// the matching bias: a projection GEMM with these rows writes the four pre-activations of a unit next to each other (16 bytes),
// which is how the persistent recurrence reads them (one 16-byte load per (row, unit) instead of four 4-byte ones).
template <typename T>
__global__ void lstm_rows_gate_minor_kernel(const T* __restrict__ src, const float* __restrict__ bias_p, int U, int ld, T* __restrict__ dst,
                                            float* __restrict__ bias_gm) {
    const int row = blockIdx.x;                      // destination row = unit * 4 + g
    const int unit = row >> 2, g = row & 3;
    const int pc = gate_perm_col(g, unit);
    for (int k = threadIdx.x; k < ld; k += blockDim.x) dst[(size_t)row * ld + k] = src[(size_t)pc * ld + k];
    if (threadIdx.x == 0) bias_gm[row] = bias_p[pc];
}

extern "C" int mnn_lstm_rows_gate_minor(mnn_stream_t s, int dtype, int units, int ld, const void* wx_t, const float* bias_p, void* wx_gm,
                                        float* bias_gm) {
    MNN_REQUIRE(wx_t && bias_p && wx_gm && bias_gm && units > 0 && units % 32 == 0 && ld > 0, "mnn_lstm_rows_gate_minor: bad arguments");
    MNN_REQUIRE(dtype == MNN_F32 || dtype == MNN_BF16 || dtype == MNN_F16, "mnn_lstm_rows_gate_minor: dtype must be f32/bf16/f16");   // 16-bit rows move as raw words
    if (dtype == MNN_F32)
        hipLaunchKernelGGL(lstm_rows_gate_minor_kernel<float>, dim3(4 * units), dim3(128), 0, (hipStream_t)s, (const float*)wx_t, bias_p, units, ld,
                           (float*)wx_gm, bias_gm);
    else
        hipLaunchKernelGGL(lstm_rows_gate_minor_kernel<bf16_t>, dim3(4 * units), dim3(128), 0, (hipStream_t)s, (const bf16_t*)wx_t, bias_p, units, ld,
                           (bf16_t*)wx_gm, bias_gm);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// dW[k][nat] += src_t[pc(nat)][k], db[nat] += db_p[pc(nat)]: a transpose with the gate permutation.  One gate of one 32-unit block is
// 32 consecutive rows pc of the packed sources AND 32 consecutive natural columns, so 32 x 32 tiles through LDS read along k and
// write along nat, both coalesced (the element-per-thread form gathered with a stride of ld_in floats).  CONSUME: every source element
// is set back to zero by the thread that read it -- the packed buffers are then persistent accumulators for the split-K GEMMs.
template <bool CONSUME>
__global__ void __launch_bounds__(256)
lstm_unpack_grads_kernel(float* __restrict__ dwx_t, float* __restrict__ dwh_t, float* __restrict__ db_p, int n_in, int U, int ld_in, int ld_h,
                         float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int pc0 = blockIdx.x * 32, ub = blockIdx.x >> 2, g = blockIdx.x & 3;
    const int nat0 = g * U + ub * 32, N4 = 4 * U, k0 = blockIdx.y * 32, K = n_in + U;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int pc = pc0 + ty + 8 * jj, k = k0 + tx;
        float val = 0.f;
        if (k < K) {
            float* src = k < n_in ? dwx_t + (size_t)pc * ld_in + k : dwh_t + (size_t)pc * ld_h + (k - n_in);
            val = *src;
            if (CONSUME) *src = 0.f;
        }
        tile[ty + 8 * jj][tx] = val;
    }
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int k = k0 + ty + 8 * jj;
        if (k < K) dW[(size_t)k * N4 + nat0 + tx] += tile[tx][ty + 8 * jj];
    }
    if (blockIdx.y == 0 && ty == 0) {
        db[nat0 + tx] += db_p[pc0 + tx];
        if (CONSUME) db_p[pc0 + tx] = 0.f;
    }
}

extern "C" int mnn_lstm_unpack_grads_consume(mnn_stream_t s, float* dwx_t, float* dwh_t, float* db_p, int n_in, int units, int ld_in, float* dW,
                                             float* db) {
    MNN_REQUIRE(dwx_t && dwh_t && db_p && dW && db && units % 32 == 0 && ld_in >= n_in, "mnn_lstm_unpack_grads_consume: bad arguments");
    dim3 grid(4 * units / 32, cdiv(n_in + units, 32));
    hipLaunchKernelGGL(lstm_unpack_grads_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, dwx_t, dwh_t, db_p, n_in, units, ld_in, units, dW, db);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// the two packed gradients side by side in one matrix dw_cat [4u, ld_in + u] (columns [0, ld_in): dWx^T, [ld_in, ld_in + u): dWh^T) -- the
// output of ONE weight-gradient GEMM over the concatenated operand [x^T ; h_prev^T]; consuming form
extern "C" int mnn_lstm_unpack_grads_cat(mnn_stream_t s, float* dw_cat, float* db_p, int n_in, int units, int ld_in, float* dW, float* db) {
    MNN_REQUIRE(dw_cat && db_p && dW && db && units % 32 == 0 && ld_in >= n_in, "mnn_lstm_unpack_grads_cat: bad arguments");
    dim3 grid(4 * units / 32, cdiv(n_in + units, 32));
    hipLaunchKernelGGL(lstm_unpack_grads_kernel<true>, grid, dim3(256), 0, (hipStream_t)s, dw_cat, dw_cat + ld_in, db_p, n_in, units, ld_in + units,
                       ld_in + units, dW, db);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

extern "C" int mnn_lstm_unpack_grads(mnn_stream_t s, const float* dwx_t, const float* dwh_t, const float* db_p, int n_in, int units,
                                     int ld_in, float* dW, float* db) {
    MNN_REQUIRE(dwx_t && dwh_t && db_p && dW && db && units % 32 == 0 && ld_in >= n_in, "mnn_lstm_unpack_grads: bad arguments");
    dim3 grid(4 * units / 32, cdiv(n_in + units, 32));
    hipLaunchKernelGGL(lstm_unpack_grads_kernel<false>, grid, dim3(256), 0, (hipStream_t)s, const_cast<float*>(dwx_t), const_cast<float*>(dwh_t),
                       const_cast<float*>(db_p), n_in, units, ld_in, units, dW, db);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// dropout (rnn.py:132 DropoutWrapper, output only): one Philox call per 4 consecutive units
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ void dropout_fwd_kernel(const T* __restrict__ h, T* __restrict__ y, int Tn, int B, int U, float kp, uint64_t seed,
                                   const int32_t* __restrict__ step_dev, uint32_t row0, int layer, int t_offset) {
    if (step_dev != nullptr) seed += (uint64_t)step_dev[0];
    const int U4 = U >> 2;
    const long total = (long)Tn * B * U4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % U4);
        const long tb = i / U4;
        const int b = (int)(tb % B), t = (int)(tb / B);
        float u[4];
        philox_uniform4(seed, MNN_STREAM_DROPOUT, row0 + (uint32_t)b, ((uint32_t)(t + t_offset) << 8) | (uint32_t)layer, (uint32_t)q, u);
        const size_t o = (size_t)tb * U + q * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float keep = floorf(kp + u[k]);
            st_from_f32<T>(y, o + k, ld_as_f32<T>(h, o + k) / kp * keep);
        }
    }
}

extern "C" int mnn_dropout_fwd(mnn_stream_t s, int dtype, const void* h, void* y, int T, int B, int units, float keep_prob, uint64_t seed,
                               const int32_t* step_dev, uint32_t row0, int layer, int t_offset) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(h && y && T > 0 && B > 0 && units > 0 && units % 4 == 0, "mnn_dropout_fwd: bad arguments");
    MNN_REQUIRE(dtype == MNN_F32 || dtype == MNN_BF16 || dtype == MNN_F16, "mnn_dropout_fwd: dtype must be f32/bf16/f16");
    MNN_REQUIRE(keep_prob > 0.f, "mnn_dropout_fwd: keep_prob must be > 0");
    const size_t bytes = (size_t)T * B * units * (dtype == MNN_F32 ? 4 : 2);
    if (keep_prob >= 1.0f) {
        if (h != y) MNN_HIP(mnn_copy_async(y, h, bytes, st));
        return MNN_OK;
    }
    const int blocks = (int)min((long)4096, ((long)T * B * units / 4 + 255) / 256);
    if (dtype == MNN_F32)
        hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)h, (float*)y, T, B, units, keep_prob, seed,
                           step_dev, row0, layer, t_offset);
    else if (dtype == MNN_F16)
        hipLaunchKernelGGL(dropout_fwd_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)h, (f16_t*)y, T, B, units, keep_prob,
                           seed, step_dev, row0, layer, t_offset);
    else
        hipLaunchKernelGGL(dropout_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)y, T, B, units, keep_prob,
                           seed, step_dev, row0, layer, t_offset);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// keep flags of DropoutWrapper for a whole sequence, mask[t,b,j] = floor(kp + u) in {0,1} (same Philox counters as
// mnn_dropout_fwd), so the fused step kernels can apply dropout with one load instead of a Philox call on the
// latency-critical chain
__global__ void dropout_mask_kernel(uint8_t* __restrict__ mask, int Tn, int B, int U, float kp, uint64_t seed, const int32_t* __restrict__ step_dev,
                                    uint32_t row0, int layer) {
    if (step_dev != nullptr) seed += (uint64_t)step_dev[0];
    const int U4 = U >> 2;
    const long total = (long)Tn * B * U4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % U4);
        const long tb = i / U4;
        const int b = (int)(tb % B), t = (int)(tb / B);
        float u[4];
        philox_uniform4(seed, MNN_STREAM_DROPOUT, row0 + (uint32_t)b, ((uint32_t)t << 8) | (uint32_t)layer, (uint32_t)q, u);
        uchar4 k;
        k.x = (uint8_t)floorf(kp + u[0]); k.y = (uint8_t)floorf(kp + u[1]); k.z = (uint8_t)floorf(kp + u[2]); k.w = (uint8_t)floorf(kp + u[3]);
        *reinterpret_cast<uchar4*>(mask + (size_t)tb * U + q * 4) = k;
    }
}

// Same flags, 16 per thread (four Philox blocks, one 16-byte store), 32-bit index arithmetic: units % 16 == 0 and fewer than 2^31 elements.
__global__ void __launch_bounds__(256)
dropout_mask16_kernel(uint8_t* __restrict__ mask, int rows, int B, int U16, float kp, uint64_t seed, const int32_t* __restrict__ step_dev,
                      uint32_t row0, int layer) {
    if (step_dev != nullptr) seed += (uint64_t)step_dev[0];
    const unsigned total = (unsigned)rows * (unsigned)U16;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned tb = i / (unsigned)U16, q16 = i - tb * (unsigned)U16;
        const unsigned t = tb / (unsigned)B, b = tb - t * (unsigned)B;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float u[4];
            philox_uniform4(seed, MNN_STREAM_DROPOUT, row0 + b, (t << 8) | (uint32_t)layer, q16 * 4u + (uint32_t)j, u);
            w[j] = (uint32_t)floorf(kp + u[0]) | ((uint32_t)floorf(kp + u[1]) << 8) | ((uint32_t)floorf(kp + u[2]) << 16) | ((uint32_t)floorf(kp + u[3]) << 24);
        }
        *reinterpret_cast<uint4*>(mask + (size_t)i * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

extern "C" int mnn_dropout_mask(mnn_stream_t s, uint8_t* mask, int T, int B, int units, float keep_prob, uint64_t seed, const int32_t* step_dev,
                                uint32_t row0, int layer) {
    MNN_REQUIRE(mask && T > 0 && B > 0 && units > 0 && units % 4 == 0 && keep_prob > 0.f && keep_prob < 1.f, "mnn_dropout_mask: bad arguments");
    const long total = (long)T * B * units;
    if (units % 16 == 0 && total < (1L << 31) && ((uintptr_t)mask & 15) == 0) {
        const int blocks = (int)min((long)8192, (total / 16 + 255) / 256);
        hipLaunchKernelGGL(dropout_mask16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, mask, T * B, B, units / 16, keep_prob, seed, step_dev, row0,
                           layer);
        MNN_LAUNCH_CHECK();
        return MNN_OK;
    }
    const int blocks = (int)min((long)4096, ((long)T * B * units / 4 + 255) / 256);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, mask, T, B, units, keep_prob, seed, step_dev, row0, layer);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dh, int Tn, int B, int U, float kp, uint64_t seed,
                                   const int32_t* __restrict__ step_dev, uint32_t row0, int layer, int accumulate, int t_offset) {
    if (step_dev != nullptr) seed += (uint64_t)step_dev[0];
    const int U4 = U >> 2;
    const long total = (long)Tn * B * U4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % U4);
        const long tb = i / U4;
        const int b = (int)(tb % B), t = (int)(tb / B);
        float u[4] = {1.f, 1.f, 1.f, 1.f};
        if (kp < 1.0f) philox_uniform4(seed, MNN_STREAM_DROPOUT, row0 + (uint32_t)b, ((uint32_t)(t + t_offset) << 8) | (uint32_t)layer, (uint32_t)q, u);
        const size_t o = (size_t)tb * U + q * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = kp < 1.0f ? dy[o + k] / kp * floorf(kp + u[k]) : dy[o + k];
            dh[o + k] = accumulate ? dh[o + k] + g : g;
        }
    }
}

extern "C" int mnn_dropout_bwd(mnn_stream_t s, const float* dy, float* dh, int T, int B, int units, float keep_prob, uint64_t seed,
                               const int32_t* step_dev, uint32_t row0, int layer, int accumulate, int t_offset) {
    MNN_REQUIRE(dy && dh && T > 0 && B > 0 && units > 0 && units % 4 == 0 && keep_prob > 0.f, "mnn_dropout_bwd: bad arguments");
    const int blocks = (int)min((long)4096, ((long)T * B * units / 4 + 255) / 256);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, dy, dh, T, B, units, keep_prob, seed, step_dev, row0,
                       layer, accumulate, t_offset);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// reductions + optimiser (utils/training.py:163-175, train.py:61-64)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_sum_256(float v) {
    __shared__ float part[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}

// few workgroups, 16-byte loads: the partial sums all land on ONE word, and same-address float atomics serialise at ~12 ns each
// (1024 of them were 12 of this kernel's 16 us)
__global__ void __launch_bounds__(256) sumsq_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
    float acc = 0.f;
    const long n4 = (((uintptr_t)x & 15) == 0) ? n / 4 : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 q = x4[i];
        acc = fmaf(q.x, q.x, acc); acc = fmaf(q.y, q.y, acc); acc = fmaf(q.z, q.z, acc); acc = fmaf(q.w, q.w, acc);
    }
    for (long i = n4 * 4 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc = fmaf(x[i], x[i], acc);
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) atomicAdd(out, acc);
}
__global__ void __launch_bounds__(256) wsum_kernel(const float* __restrict__ x, const float* __restrict__ w, long n, float* __restrict__ out) {
    float acc = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc = fmaf(x[i], w ? w[i] : 1.f, acc);
    acc = block_sum_256(acc);
    if (threadIdx.x == 0) atomicAdd(out, acc);
}

extern "C" int mnn_sumsq(mnn_stream_t s, const float* x, long n, float* out) {
    MNN_REQUIRE(x && out && n > 0, "mnn_sumsq: bad arguments");
    hipLaunchKernelGGL(sumsq_kernel, dim3((int)min(256L, (n + 1023) / 1024)), dim3(256), 0, (hipStream_t)s, x, n, out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
extern "C" int mnn_weighted_sum(mnn_stream_t s, const float* x, const float* w, long n, float* out) {
    MNN_REQUIRE(x && out && n > 0, "mnn_weighted_sum: bad arguments");
    hipLaunchKernelGGL(wsum_kernel, dim3((int)min(1024L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, w, n, out);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

__global__ void clip_adam_kernel(float* __restrict__ theta, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                                 long n, const float* __restrict__ sumsq, float clip, float lr_t, float lr, float b1, float b2, float eps,
                                 int sgd, const int32_t* __restrict__ step_dev, int32_t* __restrict__ skipped) {
    // a gradient norm that is not finite (an overflow of the loss-scaled f16 backward pass, or a NaN from anywhere): the update is NOT
    // applied -- theta, m, v keep their values -- and the caller's counter records it (read by ParamStore.check on the host)
    if (clip > 0.f && sumsq != nullptr && !isfinite(sumsq[0])) {
        if (skipped != nullptr && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(skipped, 1);
        return;
    }
    if (step_dev != nullptr) {          // step counter lives on the device (hipGraph replay): t = *step_dev + 1
        __shared__ float s_lr_t;        // two double-precision pow() once per workgroup, not once per thread
        if (threadIdx.x == 0) {
            const double t = (double)(step_dev[0] + 1);
            s_lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
        }
        __syncthreads();
        lr_t = s_lr_t;
    }
    float scale = 1.f;
    if (clip > 0.f && sumsq != nullptr) {
        const float gn = sqrtf(sumsq[0]);
        scale = gn > 0.f ? clip * fminf(1.0f / gn, 1.0f / clip) : 1.f;   // tf.clip_by_global_norm
    }
    // 16-byte accesses over the aligned body (seven streams of n floats: a memory-bound pass), scalar tail
    const bool al = ((((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const long n4 = al ? n / 4 : 0;
    if (!sgd) {
        float4* t4 = reinterpret_cast<float4*>(theta);
        const float4* g4 = reinterpret_cast<const float4*>(grad);
        float4* m4 = reinterpret_cast<float4*>(m);
        float4* v4 = reinterpret_cast<float4*>(v);
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
            const float4 gq = g4[i];
            float4 mq = m4[i], vq = v4[i], tq = t4[i];
            const float ge[4] = {gq.x * scale, gq.y * scale, gq.z * scale, gq.w * scale};
            float me[4] = {mq.x, mq.y, mq.z, mq.w}, ve[4] = {vq.x, vq.y, vq.z, vq.w}, te[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                me[e] = b1 * me[e] + (1.f - b1) * ge[e];
                ve[e] = b2 * ve[e] + (1.f - b2) * ge[e] * ge[e];
                te[e] -= lr_t * me[e] / (sqrtf(ve[e]) + eps);
            }
            m4[i] = make_float4(me[0], me[1], me[2], me[3]);
            v4[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
            t4[i] = make_float4(te[0], te[1], te[2], te[3]);
        }
    }
    for (long i = (sgd ? 0 : n4 * 4) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = grad[i] * scale;
        if (sgd) {
            theta[i] -= lr * g;
        } else {
            const float mi = b1 * m[i] + (1.f - b1) * g;
            const float vi = b2 * v[i] + (1.f - b2) * g * g;
            m[i] = mi;
            v[i] = vi;
            theta[i] -= lr_t * mi / (sqrtf(vi) + eps);
        }
    }
}

extern "C" int mnn_clip_adam_step(mnn_stream_t s, float* theta, const float* grad, float* m, float* v, long n, const float* sumsq,
                                  float clip_norm, float lr, float beta1, float beta2, float eps, int step, const int32_t* step_dev, int sgd,
                                  int32_t* skipped) {
    MNN_REQUIRE(theta && grad && n > 0 && (sgd || (m && v)) && (step >= 1 || step_dev != nullptr), "mnn_clip_adam_step: bad arguments");
    if (step < 1) step = 1;
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, step)) / (1.0 - pow((double)beta1, step));
    hipLaunchKernelGGL(clip_adam_kernel, dim3((int)min(2048L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, theta, grad, m, v, n, sumsq,
                       clip_norm, (float)lr_t, lr, beta1, beta2, eps, sgd, step_dev, skipped);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// db[c] (+)= sum_r dY[r, c]; grid over column blocks of 64 x row slabs, atomics across slabs
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ dY, int rows, int cols, int ld, float* __restrict__ db) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < cols)
        for (int r = blockIdx.y * 4 + w; r < rows; r += gridDim.y * 4) acc += dY[(size_t)r * ld + c];
    part[w][threadIdx.x & 63] = acc;
    __syncthreads();
    if (w == 0 && c < cols) atomicAdd(db + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

extern "C" int mnn_bias_grad(mnn_stream_t s, const float* dY, int rows, int cols, int ld, float* db, int accumulate) {
    hipStream_t st = (hipStream_t)s;
    MNN_REQUIRE(dY && db && rows > 0 && cols > 0 && ld >= cols, "mnn_bias_grad: bad arguments");
    if (!accumulate) MNN_HIP(mnn_zero_async(db, (size_t)cols * 4, st));
    dim3 grid(cdiv(cols, 64), min(256, cdiv(rows, 64)));
    hipLaunchKernelGGL(bias_grad_kernel, grid, dim3(256), 0, st, dY, rows, cols, ld, db);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// One pass over an f32 gradient block dY[rows, cols_c] that feeds three consumers (generators.py backward of the dense layer):
//   out_c[r, c]  = bf16(dY[r, c])            c < cols_t, 0 for cols_t <= c < cols_c   (A operand of the data-gradient GEMM, K padded)
//   out_t[c, r]  = bf16(dY[r, c])            c < cols_t   (B operand of the weight-gradient GEMM, K-major)
//   db[c]       += sum_r dY[r, c]            c < cols_t   (bias gradient)
// instead of mnn_convert2d + mnn_transpose + mnn_bias_grad, each of which re-read the f32 block.
// 64 x 64 tiles, 256 threads: thread (ty = tid / 16, tx = tid % 16) loads float4 at rows ty + 16 k, columns 4 tx.
template <typename F>
__global__ void __launch_bounds__(256)
grad_rows_fanout_kernel(const float* __restrict__ dY, int rows, int cols_c, int cols_t, int ld, bf16_t* __restrict__ out_c, int ld_c,
                        bf16_t* __restrict__ out_t, int ld_t, float* __restrict__ db) {
    __shared__ bf16_t tile[64][66];
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = blockIdx.y * 64; r0 < rows; r0 += gridDim.y * 64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            if (r < rows) {                 // columns [cols_t, cols_c) are padding: written as zeros, never read (dY's may be uninitialised)
                if (c + 3 < cols_t && (ld & 3) == 0) {
                    const float4 q = *reinterpret_cast<const float4*>(dY + (size_t)r * ld + c);
                    x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < cols_t) x[e] = dY[(size_t)r * ld + c + e];
                }
            }
            bf16_t b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b[e] = F::cvt(x[e]);
                acc[e] += x[e];
                tile[ty + 16 * k][4 * tx + e] = b[e];
            }
            if (r < rows) {
                if (c + 3 < cols_c && (ld_c & 3) == 0) {
                    uint2 pk;
                    pk.x = (uint32_t)b[0] | ((uint32_t)b[1] << 16);
                    pk.y = (uint32_t)b[2] | ((uint32_t)b[3] << 16);
                    *reinterpret_cast<uint2*>(out_c + (size_t)r * ld_c + c) = pk;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < cols_c) out_c[(size_t)r * ld_c + c + e] = b[e];
                }
            }
        }
        __syncthreads();
        {   // transposed copy: thread (column tc, row quarter rq) writes 16 consecutive rows of its column
            const int tc = threadIdx.x >> 2, rq = threadIdx.x & 3;
            const int c = c0 + tc;
            if (c < cols_t) {
                bf16_t* dst = out_t + (size_t)c * ld_t + r0 + 16 * rq;
                if (r0 + 16 * rq + 15 < rows && (ld_t & 7) == 0) {
                    uint32_t w[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = (uint32_t)tile[16 * rq + 2 * e][tc] | ((uint32_t)tile[16 * rq + 2 * e + 1][tc] << 16);
                    reinterpret_cast<uint4*>(dst)[0] = make_uint4(w[0], w[1], w[2], w[3]);
                    reinterpret_cast<uint4*>(dst)[1] = make_uint4(w[4], w[5], w[6], w[7]);
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (r0 + 16 * rq + e < rows) dst[e] = tile[16 * rq + e][tc];
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[ty][4 * tx + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < cols_t) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += part[k][threadIdx.x];
        atomicAdd(db + c0 + threadIdx.x, sum);
    }
}

extern "C" int mnn_grad_rows_fanout(mnn_stream_t s, const float* dY, int rows, int cols_c, int cols_t, int ld, void* out_c, int ld_c,
                                    void* out_t, int ld_t, float* db, int dtype) {
    MNN_REQUIRE(dtype == MNN_BF16 || dtype == MNN_F16, "mnn_grad_rows_fanout: dtype must be bf16 or f16");
    MNN_REQUIRE(dY && out_c && out_t && db && rows > 0 && cols_c > 0 && cols_t > 0 && cols_t <= cols_c, "mnn_grad_rows_fanout: bad arguments");
    MNN_REQUIRE(ld >= cols_c && ld_c >= cols_c && ld_t >= rows, "mnn_grad_rows_fanout: leading dimension too small");
    MNN_REQUIRE(((uintptr_t)dY & 15) == 0 && ((uintptr_t)out_c & 7) == 0 && ((uintptr_t)out_t & 15) == 0, "mnn_grad_rows_fanout: misaligned buffer");
    dim3 grid(cdiv(cols_c, 64), min(256, cdiv(rows, 64)));
    if (dtype == MNN_F16)
        hipLaunchKernelGGL(grad_rows_fanout_kernel<Fp16F>, grid, dim3(256), 0, (hipStream_t)s, dY, rows, cols_c, cols_t, ld, (bf16_t*)out_c, ld_c,
                           (bf16_t*)out_t, ld_t, db);
    else
        hipLaunchKernelGGL(grad_rows_fanout_kernel<Bf16F>, grid, dim3(256), 0, (hipStream_t)s, dY, rows, cols_c, cols_t, ld, (bf16_t*)out_c, ld_c,
                           (bf16_t*)out_t, ld_t, db);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// the same test as clip_adam_kernel: a step whose gradient norm is not finite was NOT applied, so it does not count (the Adam bias
// correction of the next applied step uses the number of APPLIED steps)
// ... and the f16 loss scale follows the outcome (ls_dyn = [m, 1 / m], the power-of-two multiplier the owner of a backward pass applies on top of
// its static scale): a skipped step halves m (the backward overflowed -- with a static scale every later step would overflow too and training
// stands still for good: seen at the bench shape after ~270 steps at lr 0.01), `grow_after` applied steps in a row double it back, up to 1.
__global__ void step_increment_kernel(int32_t* step_dev, const float* __restrict__ sumsq, float clip, float* ls_dyn, int32_t* ls_good,
                                      int grow_after, float m_min) {
    const bool skippedstep = clip > 0.f && sumsq != nullptr && !isfinite(sumsq[0]);
    if (ls_dyn != nullptr) {
        float m = ls_dyn[0];
        if (skippedstep) {
            m = fmaxf(m * 0.5f, m_min);
            ls_good[0] = 0;
        } else if (++ls_good[0] >= grow_after && m < 1.0f) {
            m = fminf(m * 2.0f, 1.0f);
            ls_good[0] = 0;
        }
        ls_dyn[0] = m;
        ls_dyn[1] = 1.0f / m;
    }
    if (skippedstep) return;
    step_dev[0] += 1;
}
extern "C" int mnn_step_increment(mnn_stream_t s, int32_t* step_dev, const float* sumsq, float clip_norm, float* ls_dyn, int32_t* ls_good,
                                  int grow_after) {
    MNN_REQUIRE(step_dev, "mnn_step_increment: null pointer");
    MNN_REQUIRE((ls_dyn == nullptr) == (ls_good == nullptr) && (ls_dyn == nullptr || grow_after > 0), "mnn_step_increment: ls_dyn [2] and ls_good [1] come together, grow_after > 0");
    hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, step_dev, sumsq, clip_norm, ls_dyn, ls_good, grow_after,
                       1.0f / 1048576.0f);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

__global__ void fill_kernel(float* __restrict__ x, long n, float v) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = v;
}
extern "C" int mnn_fill_f32(mnn_stream_t s, float* x, long n, float value) {
    MNN_REQUIRE(x && n > 0, "mnn_fill_f32: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3((int)min(2048L, (n + 255) / 256)), dim3(256), 0, (hipStream_t)s, x, n, value);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// Measurement support ("the builder must measure the device's sigmoid (v_exp_f32 + v_rcp_f32) throughput with a
// microbenchmark and use it as [the NADE] phase's peak").  Every thread runs 8 independent chains of the NADE kernels' own sigmoid
// (fast_sigmoid: v_mul, v_exp_f32, v_add, v_rcp_f32), `iters` links each, 8 waves per SIMD resident; nothing touches memory until
// the final store.  bench.py times the launch with HIP events: sigmoids/s = grid * 256 * 8 * iters / t.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) probe_sigmoid_kernel(float* __restrict__ out, int iters) {
    float x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = 1e-3f * (float)threadIdx.x + 0.125f * (float)c - 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = fast_sigmoid(x[c]);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int mnn_probe_sigmoid(mnn_stream_t s, int blocks, int iters, float* out) {
    MNN_REQUIRE(blocks > 0 && iters > 0 && out, "mnn_probe_sigmoid: bad arguments");
    hipLaunchKernelGGL(probe_sigmoid_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, out, iters);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}

// ----------------------------------------------------------------------------------------------
// Density gate: gate[0] = (number of non-zero bytes of v[0..n) > threshold) ? 1 : 0, decided ON THE DEVICE so that a captured step picks
// per replay between the two forms of the NADE forward scan (matrix-core block-sparse form: cost grows with the number of active
// visibles; f32 vector form: nearly flat) -- see mnn_nade_logprob_fwd_gated.  Two launches: count (one u32 atomic per workgroup),
// decide.  count[0] must be zero on entry; the decide kernel re-zeroes it.
// ----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) density_count_kernel(const uint8_t* __restrict__ v, long n, unsigned* __restrict__ count) {
    unsigned acc = 0;
    const long n16 = (((uintptr_t)v & 15) == 0) ? n / 16 : 0;
    const uint4* v16 = reinterpret_cast<const uint4*>(v);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const uint4 q = v16[i];
        // bytes are 0 / 1 piano-roll cells (any non-zero value counts once): fold each byte to its lowest bit, then popcount
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned x = w[k];
            x |= x >> 4; x |= x >> 2; x |= x >> 1;
            acc += __popc(x & 0x01010101u);
        }
    }
    for (long i = n16 * 16 + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += v[i] != 0;
    __shared__ unsigned part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count + (blockIdx.x & (MNN_DENSITY_SLOTS - 1)), part[0] + part[1] + part[2] + part[3]);
}
__global__ void __launch_bounds__(MNN_DENSITY_SLOTS) density_decide_kernel(unsigned* __restrict__ count, unsigned long long threshold, int* __restrict__ gate) {
    __shared__ unsigned long long tot;
    if (threadIdx.x == 0) tot = 0ull;
    __syncthreads();
    const unsigned c = count[threadIdx.x];                // the partial counts of the MNN_DENSITY_SLOTS slots; left zero for the next pass
    count[threadIdx.x] = 0u;
    if (c != 0u) atomicAdd(&tot, (unsigned long long)c);
    __syncthreads();
    if (threadIdx.x == 0) gate[0] = tot > threshold ? 1 : 0;
}
extern "C" int mnn_density_gate(mnn_stream_t s, const uint8_t* v, long n, long threshold, int* gate, unsigned* count) {
    MNN_REQUIRE(gate && count && n > 0 && threshold >= 0 && n < (1L << 32), "mnn_density_gate: bad arguments");
    // v == NULL: *count already holds the number of set cells (mnn_pianoroll_shift_timemajor_t counted them while writing the targets)
    if (v != nullptr) hipLaunchKernelGGL(density_count_kernel, dim3((int)min(1024L, (n + 4095) / 4096)), dim3(256), 0, (hipStream_t)s, v, n, count);
    hipLaunchKernelGGL(density_decide_kernel, dim3(1), dim3(MNN_DENSITY_SLOTS), 0, (hipStream_t)s, count, (unsigned long long)threshold, gate);
    MNN_LAUNCH_CHECK();
    return MNN_OK;
}
