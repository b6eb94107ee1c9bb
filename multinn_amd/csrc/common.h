// Shared device/host helpers for libmultinn_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/multinn_hip.h"

// ----------------------------------------------------------------------------------------------
// error handling: every C-ABI entry returns 0 or <0 and records a thread-local message
// ----------------------------------------------------------------------------------------------
void mnn_set_error(const char* fmt, ...);

#define MNN_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            mnn_set_error(__VA_ARGS__);        \
            return MNN_ERR_INVALID;            \
        }                                      \
    } while (0)

#define MNN_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            mnn_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return MNN_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define MNN_LAUNCH_CHECK() MNN_HIP(hipGetLastError())

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// "done once" flags of things that are set PER DEVICE (hipFuncSetAttribute: a process-wide flag would leave the second device of a process at
// the default dynamic-LDS limit): the current device's slot of a static array
static inline bool& mnn_dev_flag(bool (&flags)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return flags[dev];
}

// ----------------------------------------------------------------------------------------------
// Stream-ordered zero fills and device copies as plain kernels.  The library never calls hipMemsetAsync / hipMemcpyAsync
// on a compute stream: their hipGraph memset nodes were observed (ROCm 7.2, gfx950) to fill with a stale 16-byte pattern
// on the second replay of a graph once another graph had been instantiated in the process (the data-parallel step replays
// two graphs around its all-reduce) -- the hand-off flags of the persistent recurrence then start non-zero.  A kernel node
// carries its own arguments.  Templates so that every translation unit instantiates only what it uses (no RDC).
// ----------------------------------------------------------------------------------------------
template <int UNUSED>
__global__ void __launch_bounds__(256) mnn_zero_rows_kernel(char* p, size_t row_bytes, size_t pitch, int rows) {
    // row_bytes and pitch multiples of 4, p 4-byte aligned; 16-byte stores where the row allows
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        char* row = p + (size_t)r * pitch;
        const size_t head = (16 - ((uintptr_t)row & 15)) & 15;                  // bytes up to the first 16-byte boundary
        const size_t h4 = (head < row_bytes ? head : row_bytes) / 4;
        const size_t n16 = (row_bytes - h4 * 4) / 16, t4 = (row_bytes - h4 * 4 - n16 * 16) / 4;
        const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
        if (tid < h4) reinterpret_cast<uint32_t*>(row)[tid] = 0u;
        uint4* v = reinterpret_cast<uint4*>(row + h4 * 4);
        for (size_t i = tid; i < n16; i += nth) v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < t4) reinterpret_cast<uint32_t*>(row + h4 * 4 + n16 * 16)[tid] = 0u;
    }
}
template <int UNUSED>
__global__ void __launch_bounds__(256) mnn_copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// zero `rows` rows of `row_bytes` bytes, `pitch` bytes apart (rows = 1: a flat fill)
static inline hipError_t mnn_zero_async(void* p, size_t row_bytes, size_t pitch, int rows, hipStream_t st) {
    if (row_bytes == 0 || rows <= 0) return hipSuccess;
    if (((uintptr_t)p & 3) || (row_bytes & 3) || (pitch & 3)) return hipErrorInvalidValue;
    const int gx = (int)((row_bytes / 16 + 255) / 256);
    dim3 grid(gx < 1 ? 1 : (gx > 1024 ? 1024 : gx), rows > 1024 ? 1024 : rows);
    hipLaunchKernelGGL(mnn_zero_rows_kernel<0>, grid, dim3(256), 0, st, (char*)p, row_bytes, pitch, rows);
    return hipGetLastError();
}
static inline hipError_t mnn_zero_async(void* p, size_t bytes, hipStream_t st) { return mnn_zero_async(p, bytes, bytes, 1, st); }
static inline hipError_t mnn_copy_async(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (((uintptr_t)dst & 3) || ((uintptr_t)src & 3) || (bytes & 3)) return hipErrorInvalidValue;
    const size_t n = bytes / 4;
    const int g = (int)((n + 255) / 256);
    hipLaunchKernelGGL(mnn_copy_words_kernel<0>, dim3(g > 2048 ? 2048 : g), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst, n);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------------------
// bf16 <-> f32 (round-to-nearest-even via the hardware cast; NaN stays NaN)
// ----------------------------------------------------------------------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

// IEEE half: the second 16-bit operand flavour (precision "fp16": 11 significant bits instead of bf16's 8 at the same MFMA rate and the
// same bytes; the operands of this path -- {0,1} inputs, |h| <= 1, |w| < 1, loss-scaled gradients -- sit inside its range).  Kernels templated
// on an element type take f16_t; kernels that move raw 16-bit words (h16_t) take a flavour struct (Bf16F / Fp16F) for the conversions and
// the matrix-core instruction.
typedef _Float16 f16_t;
typedef uint16_t h16_t;
__device__ __forceinline__ float f16_to_f32(h16_t x) { return (float)__builtin_bit_cast(_Float16, x); }
__device__ __forceinline__ h16_t f32_to_f16(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }    // v_cvt_f16_f32: round to nearest even

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    __device__ static __forceinline__ float load(float x) { return x; }
    __device__ static __forceinline__ float store(float x) { return x; }
};
template <> struct Cvt<bf16_t> {
    __device__ static __forceinline__ float load(bf16_t x) { return bf16_to_f32(x); }
    __device__ static __forceinline__ bf16_t store(float x) { return f32_to_bf16(x); }
};
template <> struct Cvt<f16_t> {
    __device__ static __forceinline__ float load(f16_t x) { return (float)x; }
    __device__ static __forceinline__ f16_t store(float x) { return (f16_t)x; }
};

typedef float mnn_f32x16 __attribute__((ext_vector_type(16)));
typedef float mnn_f32x4 __attribute__((ext_vector_type(4)));
struct Bf16F {
    typedef __bf16 x8 __attribute__((ext_vector_type(8)));
    typedef bf16_t elem;
    static constexpr int DTYPE = MNN_BF16;
    __device__ static __forceinline__ float f32(h16_t x) { return bf16_to_f32(x); }
    __device__ static __forceinline__ h16_t cvt(float f) { return f32_to_bf16(f); }
    __device__ static __forceinline__ float lo(uint32_t v) { return __uint_as_float(v << 16); }            // low / high half of a packed pair
    __device__ static __forceinline__ float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
    __device__ static __forceinline__ mnn_f32x16 mfma32(x8 a, x8 b, mnn_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ mnn_f32x4 mfma16(x8 a, x8 b, mnn_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
struct Fp16F {
    typedef _Float16 x8 __attribute__((ext_vector_type(8)));
    typedef f16_t elem;
    static constexpr int DTYPE = MNN_F16;
    __device__ static __forceinline__ float f32(h16_t x) { return f16_to_f32(x); }
    __device__ static __forceinline__ h16_t cvt(float f) { return f32_to_f16(f); }
    __device__ static __forceinline__ float lo(uint32_t v) { return f16_to_f32((h16_t)(v & 0xffffu)); }
    __device__ static __forceinline__ float hi(uint32_t v) { return f16_to_f32((h16_t)(v >> 16)); }
    __device__ static __forceinline__ mnn_f32x16 mfma32(x8 a, x8 b, mnn_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ mnn_f32x4 mfma16(x8 a, x8 b, mnn_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename T> struct FlavorOf;
template <> struct FlavorOf<bf16_t> { typedef Bf16F type; };
template <> struct FlavorOf<f16_t> { typedef Fp16F type; };
template <typename F> __device__ __forceinline__ uint32_t pack2(float a, float b) { return (uint32_t)F::cvt(a) | ((uint32_t)F::cvt(b) << 16); }

// ----------------------------------------------------------------------------------------------
// fast math (throughput paths; tolerance-checked against the oracle)
// ----------------------------------------------------------------------------------------------
#define MNN_LOG2E 1.4426950408889634f
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp2(-MNN_LOG2E * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * fast_rcp(1.0f + fast_exp2(-2.0f * MNN_LOG2E * x)) - 1.0f; }

// ----------------------------------------------------------------------------------------------
// deterministic math (sampling paths; bit-identical to oracle/det_ref.c, which restates the same
// specification with IEEE fma/mul/add/div only).  DESIGN.md "Deterministic sigmoid".
// ----------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float det_exp(float x) {
    // exp(x) for x in [-87, 87]: n = floor(x*log2e + 0.5); r = x - n*ln2 (two-term); degree-6 Horner; scale by 2^n.
    x = fminf(fmaxf(x, -87.0f), 87.0f);       // two instructions on the device; the same value as the two selects for every non-NaN x
    const float n = floorf(fmaf(x, 1.4426950408889634f, 0.5f));
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.3888889225e-3f;                 // 1/720
    p = fmaf(p, r, 8.3333337680e-3f);           // 1/120
    p = fmaf(p, r, 4.1666667908e-2f);           // 1/24
    p = fmaf(p, r, 1.6666667163e-1f);           // 1/6
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    const int e = (int)n + 127;                 // 40..213 -> normal
    union { uint32_t u; float f; } s;
    s.u = ((uint32_t)e) << 23;
    return p * s.f;
}
__host__ __device__ __forceinline__ float det_sigmoid(float x) { return 1.0f / (1.0f + det_exp(-x)); }

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 (RNG contract: DESIGN.md; oracle/philox.py)
// ----------------------------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64-bit multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 + v_mul_lo_u32 pair: the integer multiplies are
        // quarter-rate instructions and most of a Philox block's cost
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ float bits_to_uniform(uint32_t x) { return __uint_as_float((x & 0x7FFFFFu) | 0x3F800000u) - 1.0f; }

// four uniforms for elements 4q..4q+3 of (row, sub) on `stream`
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint32_t stream, uint32_t row, uint32_t sub, uint32_t q, float u[4]) {
    const Philox4 r = philox4x32_10(q, row, sub, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    u[0] = bits_to_uniform(r.x); u[1] = bits_to_uniform(r.y); u[2] = bits_to_uniform(r.z); u[3] = bits_to_uniform(r.w);
}
__device__ __forceinline__ float philox_uniform1(uint64_t seed, uint32_t stream, uint32_t row, uint32_t sub, uint32_t elem) {
    const Philox4 r = philox4x32_10(elem >> 2, row, sub, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w = elem & 3u;
    return bits_to_uniform(w == 0 ? r.x : (w == 1 ? r.y : (w == 2 ? r.z : r.w)));
}

// gate-interleaved column index of the LSTM pre-activation space (DESIGN.md "LSTM layout"):
// natural TF column g*u + unit  ->  (unit/32)*128 + g*32 + unit%32
__host__ __device__ __forceinline__ int gate_perm_col(int g, int unit) { return (unit >> 5) * 128 + g * 32 + (unit & 31); }
