// Phase clocks of ONE workgroup of gemm_tn_pair_kernel (wave 0, thread 0), accumulated over its K loop, in microseconds.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 profiles/tools/gemm_pair_trace.hip -o scratch/gemm_pair_trace -ldl && scratch/gemm_pair_trace [M N K c16 var]
#ifndef GM_TRACE
#define GM_TRACE 9000
#endif
#include "../../multinn_amd/csrc/gemm.hip"
#include <cstdarg>
#include <cstdlib>
extern "C" int mnn_transpose(mnn_stream_t, const void*, int, int, int, int, void*, int, int) { return 0; }
extern "C" int mnn_bias_grad(mnn_stream_t, const float*, int, int, int, float*, int) { return 0; }
void mnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 448, c16 = argc > 4 ? atoi(argv[4]) : 1;
    if (argc > 5) setenv("MNN_GEMM_PAIR_VAR", argv[5], 1);
    setenv("MNN_GEMM_PAIR", "1", 1);
    uint16_t *A, *B; void* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    // random f16 operands in (-1, 1): exponent field 0x30..0x3b, random mantissa and sign (zeros would clock higher)
    { std::vector<uint16_t> h((size_t)M * K); unsigned x = 12345u; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(((x >> 16) & 0x83ff) | (0x3000 + (((x >> 8) & 7) << 10))); }
      CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice)); h.resize((size_t)N * K); for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(((x >> 16) & 0x83ff) | (0x3000 + (((x >> 8) & 7) << 10))); }
      CK(hipMemcpy(B, h.data(), h.size() * 2, hipMemcpyHostToDevice)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        long long z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(gm_trace), z, sizeof(z)));
        CK(hipEventRecord(e0));
        if (mnn_gemm_tn(nullptr, MNN_F16, M, N, K, A, K, B, K, C, N, c16 ? MNN_F16 : MNN_F32, nullptr, 0, 1)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(gm_trace), sizeof(z)));
        double tot = 0; for (int k = 1; k <= 8; ++k) tot += z[k] * 0.01;
        printf("gemm %.1f us; workgroup %d (us): init + first stages %.2f | vmcnt wait %.2f | barrier %.2f | DMA issue %.2f | reads + MFMA %.2f | last barrier %.2f | epilogue issue %.2f | store drain %.2f | sum %.2f\n",
               ms * 1e3, GM_TRACE, z[1] * 0.01, z[2] * 0.01, z[3] * 0.01, z[4] * 0.01, z[5] * 0.01, z[6] * 0.01, z[7] * 0.01, z[8] * 0.01, tot);
    }
    return 0;
}
