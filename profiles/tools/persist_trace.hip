// Development harness: runs the persistent recurrence on C2-shaped buffers with per-stage timestamps (PST_TRACE).
#define PST_TRACE 1
#include "../../multinn_amd/csrc/lstm_persist.hip"
#include <cstdarg>
#include <cstdlib>
#include <vector>
void mnn_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <typename T> T* dalloc(size_t n, int fill = 0) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, fill, n * sizeof(T))); return p; }

static void report(const char* name, int role, int T) {
    static long long h[4][512][8];
    CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(pst_trace), sizeof(h)));
    double seg[6] = {0, 0, 0, 0, 0, 0};
    int n = 0;
    for (int t = 8; t < T - 1; ++t, ++n) {
        for (int k = 0; k < 5; ++k) if (h[role][t][k + 1] && h[role][t][k]) seg[k] += (double)(h[role][t][k + 1] - h[role][t][k]) * 0.01;
        seg[5] += (double)(h[role][t + 1][0] - h[role][t][0]) * 0.01;
    }
    { double cyc = 0, us = 0; for (int t = 8; t < T - 1; ++t) { cyc += (double)(h[role][t][7] - h[role][t][6]); us += (double)(h[role][t][2] - h[role][t][1]) * 0.01; }
      printf("   [%s] segment 1->2: %.0f shader cycles per %.2f us = %.2f GHz\n", name, cyc / (T - 9), us / (T - 9), cyc / us / 1000.0); }
    printf("%-8s step %.2f us: prefetch->waited %.2f | loads+mfma+reduce %.2f | pointwise+lds %.2f | stores+drain+flag %.2f | tail stores %.2f\n", name,
           seg[5] / n, seg[0] / n, seg[1] / n, seg[2] / n, seg[3] / n, seg[4] / n);
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 128, B = argc > 2 ? atoi(argv[2]) : 256, U1 = 512, U2 = 256;
    const float kp = 0.9f;
    const size_t n1 = (size_t)T * B * U1, n2 = (size_t)T * B * U2;
    const int ld = (T * B + 63) / 64 * 64;
    mnn_lstm_fwd_layer f1{}, f2{};
    f1.units = U1; f1.xproj = dalloc<float>(4 * n1); f1.wh_t = dalloc<bf16_t>((size_t)4 * U1 * U1); f1.gates = dalloc<float>(4 * n1); f1.c = dalloc<float>(n1);
    f1.h = dalloc<bf16_t>(n1); f1.hT = dalloc<bf16_t>((size_t)U1 * ld); f1.ld_hT = ld; f1.y = dalloc<bf16_t>(n1); f1.mask = dalloc<uint8_t>(n1, 1);
    f2.units = U2; f2.wh_t = dalloc<bf16_t>((size_t)4 * U2 * U2); f2.gates = dalloc<float>(4 * n2); f2.c = dalloc<float>(n2);
    f2.h = dalloc<bf16_t>(n2); f2.hT = dalloc<bf16_t>((size_t)U2 * ld); f2.ld_hT = ld; f2.y = dalloc<bf16_t>(n2); f2.mask = dalloc<uint8_t>(n2, 1);
    f2.wx_t = dalloc<bf16_t>((size_t)4 * U2 * U1); f2.ld_w = U1; f2.bias_p = dalloc<float>(4 * U2);
    void* sync = dalloc<char>(mnn_lstm2_persist_workspace_bytes(T, B, U1, U2));
    mnn_lstm_bwd_layer b1{}, b2{};
    b1.units = U1; b1.wh_p = dalloc<bf16_t>((size_t)4 * U1 * U1); b1.gates = f1.gates; b1.c = f1.c; b1.workspace = dalloc<float>((size_t)B * U1);
    b1.dzT_t = dalloc<bf16_t>((size_t)4 * U1 * ld); b1.ld_t = ld; b1.db_p = dalloc<float>(4 * U1); b1.mask = f1.mask;
    b2.units = U2; b2.dh_ext = dalloc<float>(n2); b2.wh_p = dalloc<bf16_t>((size_t)4 * U2 * U2); b2.gates = f2.gates; b2.c = f2.c; 
    b2.workspace = dalloc<float>((size_t)B * U2); b2.dzT_t = dalloc<bf16_t>((size_t)4 * U2 * ld); b2.ld_t = ld; b2.db_p = dalloc<float>(4 * U2);
    b2.wx_p = dalloc<bf16_t>((size_t)U1 * 4 * U2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0));
        if (mnn_lstm2_persist_fwd(nullptr, T, B, &f1, &f2, kp, sync)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fwd %.3f ms (%.2f us/step)\n", ms, ms * 1000 / T);
        if (rep == 2) { report("fwd L1", 0, T); report("fwd L2", 1, T); }
        CK(hipEventRecord(e0));
        if (mnn_lstm2_persist_bwd(nullptr, T, B, &b1, &b2, kp, sync)) return 1;
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("bwd %.3f ms (%.2f us/step)\n", ms, ms * 1000 / T);
        if (rep == 2) { report("bwd L2", 2, T); report("bwd L1", 3, T); }
    }
    unsigned st[1]; CK(hipMemcpy(st, sync, 4, hipMemcpyDeviceToHost));
    printf("status %u\n", st[0]);
    return 0;
}
