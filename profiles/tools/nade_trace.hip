#ifndef NOTRACE
#define NM_TRACE 100
#endif
#include "../../multinn_amd/csrc/nade_mfma.hip"
#include <cstdarg>
#include <cstdlib>
#include <vector>
void mnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char** argv) {
    const int N = (argc > 1 ? atoi(argv[1]) : 32768), D = 440, Hn = 256, ld = Hn + D;
    std::vector<uint8_t> hv((size_t)N * D);
    srand(1);
    for (auto& x : hv) x = (rand() % 1000) < 30;
    uint8_t* v; CK(hipMalloc(&v, hv.size())); CK(hipMemcpy(v, hv.data(), hv.size(), hipMemcpyHostToDevice));
    float *bias, *we, *nll, *cp, *db, *af; bf16_t* wd;
    CK(hipMalloc(&bias, (size_t)N * ld * 4)); CK(hipMemset(bias, 0, (size_t)N * ld * 4));
    CK(hipMalloc(&we, (size_t)D * Hn * 4)); CK(hipMemset(we, 0, (size_t)D * Hn * 4));
    CK(hipMalloc(&wd, (size_t)D * Hn * 2)); CK(hipMemset(wd, 0, (size_t)D * Hn * 2));
    CK(hipMalloc(&nll, N * 4)); CK(hipMalloc(&cp, (size_t)N * D * 4)); CK(hipMalloc(&db, (size_t)N * ld * 4)); CK(hipMalloc(&af, (size_t)N * Hn * 4));
    float* rw; CK(hipMalloc(&rw, N * 4)); CK(hipMemset(rw, 0, N * 4));
    float* wdp; CK(hipMalloc(&wdp, (size_t)D * Hn * 4)); CK(hipMemset(wdp, 0, (size_t)D * Hn * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        long long z[16] = {0};
#ifndef NOTRACE
        CK(hipMemcpyToSymbol(HIP_SYMBOL(nm_trace), z, sizeof(z)));
#endif
        CK(hipEventRecord(e0));
#ifdef SPLIT_FORM      // the fp16 mode's split-operand form (f16 hi + lo decoder weights, 4 bytes per weight, zeros here)
        if (mnn_nade_logprob_fwd_mfma_f32(nullptr, 1, N, D, Hn, v, (long)N * D, bias, ld, we, wdp, rw, nll, cp, db, af, nullptr, 0)) return 1;
#else
        if (mnn_nade_logprob_fwd_mfma(nullptr, 1, N, D, Hn, v, (long)N * D, bias, ld, we, wd, rw, nll, cp, db, af)) return 1;
#endif
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifndef NOTRACE
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(nm_trace), sizeof(z)));
#endif
        printf("fwd %.3f ms; block phases (us):", ms);
        const char* names[9] = {"S0 base logits + ballots", "B1", "S1 flips", "list (wave 0)", "B2", "S2 flip logits + w_enc requests", "B3", "select + w_dec requests", "S3 pointwise + stores"};
        for (int k = 0; k < 9; ++k) printf(" [%s]=%.1f", names[k], z[k] * 0.01);
        printf("\n");
    }
    return 0;
}
