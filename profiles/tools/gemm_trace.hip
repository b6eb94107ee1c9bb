#define GM_TRACE 300
#include "../../multinn_amd/csrc/gemm.hip"
#include <cstdarg>
#include <cstdlib>
extern "C" int mnn_transpose(mnn_stream_t, const void*, int, int, int, int, void*, int, int) { return 0; }
extern "C" int mnn_bias_grad(mnn_stream_t, const float*, int, int, int, float*, int) { return 0; }
void mnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 2048, K = argc > 3 ? atoi(argv[3]) : 448, sk = argc > 4 ? atoi(argv[4]) : 1;
    bf16_t *A, *B; float* C;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
    CK(hipMemset(A, 0, (size_t)M * K * 2)); CK(hipMemset(B, 0, (size_t)N * K * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        long long z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(gm_trace), z, sizeof(z)));
        CK(hipEventRecord(e0));
        if (mnn_gemm_tn(nullptr, MNN_BF16, M, N, K, A, K, B, K, C, N, MNN_F32, nullptr, 0, sk)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(gm_trace), sizeof(z)));
        printf("gemm %.1f us; workgroup %d of the 256 x 256 kernel (us): first stage+sync %.2f | issue next stages %.2f | LDS reads + MFMA %.2f | barriers %.2f | epilogue issue %.2f | stores drain %.2f\n",
               ms * 1e3, GM_TRACE, z[1] * 0.01, z[2] * 0.01, z[3] * 0.01, z[4] * 0.01, z[5] * 0.01, z[6] * 0.01);
    }
    return 0;
}
