#define NB_TRACE 100
#ifndef NB_TRACE_TID
#define NB_TRACE_TID 0
#endif
#include "../../multinn_amd/csrc/nade.hip"
#include <cstdarg>
#include <cstdlib>
#include <vector>
void mnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int N = 32768, D = 440, Hn = 256, ld = Hn + D;
    std::vector<uint8_t> hv((size_t)N * D);
    srand(1);
    for (auto& x : hv) x = (rand() % 1000) < 30;
    uint8_t* v; CK(hipMalloc(&v, hv.size())); CK(hipMemcpy(v, hv.data(), hv.size(), hipMemcpyHostToDevice));
    float *bias, *we, *wd, *db, *af, *dwe, *dwd;
    CK(hipMalloc(&bias, (size_t)N * ld * 4)); CK(hipMemset(bias, 0, (size_t)N * ld * 4));
    CK(hipMalloc(&we, (size_t)D * Hn * 4)); CK(hipMemset(we, 0, (size_t)D * Hn * 4));
    CK(hipMalloc(&wd, (size_t)D * Hn * 4)); CK(hipMemset(wd, 0, (size_t)D * Hn * 4));
    CK(hipMalloc(&dwe, (size_t)D * Hn * 4)); CK(hipMalloc(&dwd, (size_t)D * Hn * 4));
    CK(hipMalloc(&db, (size_t)N * ld * 4)); CK(hipMemset(db, 0, (size_t)N * ld * 4)); CK(hipMalloc(&af, (size_t)N * Hn * 4)); CK(hipMemset(af, 0, (size_t)N * Hn * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        long long z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(nb_trace), z, sizeof(z)));
        CK(hipEventRecord(e0));
        if (mnn_nade_logprob_bwd(nullptr, 1, N, D, Hn, v, (long)N * D, bias, ld, we, wd, af, db, dwe, dwd, nullptr, nullptr)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(nb_trace), sizeof(z)));
        const char* names[9] = {"loop top", "prefetch+ballot", "compute 4 vis", "B1 wait", "red write", "B2 wait", "reduce+atomics", "lstore", "B3 wait"};
        printf("bwd %.3f ms; wave phases (us):", ms);
        for (int k = 0; k < 9; ++k) printf(" %s=%.1f", names[k], z[k] * 0.01);
        printf("\n");
    }
    return 0;
}
