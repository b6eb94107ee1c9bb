// probe: v_mfma_f32_4x4x4_16B_f16 operand layout + issue cost on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const _Float16* A, const _Float16* B, float* D) {
    // A: [16 blocks][4 i][4 k], B: [16 blocks][4 k][4 j], D: [16 blocks][4 i][4 j]
    const int l = threadIdx.x, b = l >> 2, x = l & 3;
    f16x4 a, bb;
    for (int k = 0; k < 4; ++k) { a[k] = A[(b * 4 + x) * 4 + k]; bb[k] = B[(b * 4 + k) * 4 + x]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(b * 4 + i) * 4 + x] = c[i];
}

template <int MODE>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, int iters) {
    f16x4 a = {(_Float16)1.f, (_Float16)0.5f, (_Float16)0.25f, (_Float16)0.125f}, b = a;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c3, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
            v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
            v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
        }
        if (MODE == 3) {     // one accumulator: dependent chain
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c0, 0, 0, 0);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / iters;
}

int main() {
    std::vector<_Float16> hA(256), hB(256);
    for (int i = 0; i < 256; ++i) { hA[i] = (_Float16)((i * 7 % 13) - 6); hB[i] = (_Float16)((i * 5 % 11) - 5); }
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 512, hipMemcpyHostToDevice);
    layout_kernel<<<1, 64>>>(dA, dB, dD);
    std::vector<float> hD(256);
    hipMemcpy(hD.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float r = 0; for (int k = 0; k < 4; ++k) r += (float)hA[(b * 4 + i) * 4 + k] * (float)hB[(b * 4 + k) * 4 + j];
        if (r != hD[(b * 4 + i) * 4 + j]) ++bad;
    }
    printf("layout check (A lane=4b+i holds k, B lane=4b+j holds k, D lane=4b+j regs i): %d mismatches of 256\n", bad);
    float* out; hipMalloc(&out, ((1 << 20) + 16) * 4);
    const char* names[4] = {"4 mfma 4x4x4 (4 accumulators)", "8 v_fma", "4 mfma + 8 v_fma", "4 mfma, one accumulator"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int mode = 0; mode < 4; ++mode) for (int waves = 1; waves <= 4; waves *= 2) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate_kernel<0><<<256, 256 * waves>>>(out, iters);
            if (mode == 1) rate_kernel<1><<<256, 256 * waves>>>(out, iters);
            if (mode == 2) rate_kernel<2><<<256, 256 * waves>>>(out, iters);
            if (mode == 3) rate_kernel<3><<<256, 256 * waves>>>(out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-34s %d wave(s)/SIMD: %.3f ms -> %.1f ns per iteration per wave-slot, %.2f ns per iteration / waves\n", names[mode], waves, ms, ms * 1e6 / iters, ms * 1e6 / iters / waves);
    }
    return 0;
}
