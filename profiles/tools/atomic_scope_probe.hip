// f32 atomic adds into a per-XCD copy of a table: agent scope (the default atomicAdd) against workgroup scope (no sc1: may execute in the XCD's L2).
// Every workgroup adds 1.0 to every element of rows of ITS XCD's copy (XCC_ID from the hardware register): all adders of a copy share one L2,
// so the sums must come out exact in both forms.   hipcc -O3 --offload-arch=gfx950 profiles/tools/atomic_scope_probe.hip -o scratch/atomic_scope_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SCOPE>   // 0 agent, 1 workgroup, 2 plain store (reference rate)
__global__ void __launch_bounds__(256) add_kernel(float* __restrict__ tab, int rows, int cols, int reps, unsigned* __restrict__ xcd_count) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    if (threadIdx.x == 0) atomicAdd(xcd_count + xcc, 1u);
    float* copy = tab + (size_t)xcc * rows * cols;
    for (int r = 0; r < reps; ++r) {
        const int row = (blockIdx.x * 7 + r * 13) % rows;
        for (int c = threadIdx.x; c < cols; c += 256) {
            float* p = copy + (size_t)row * cols + c;
            if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *p = 1.0f;
        }
    }
}

int main() {
    const int rows = 440, cols = 512, reps = 64, nwg = 4096;      // one copy = 440 x 512 f32 = 0.9 MB (the NADE weight-gradient pair), 8 copies
    float* tab; unsigned* cnt;
    CK(hipMalloc(&tab, (size_t)8 * rows * cols * 4)); CK(hipMalloc(&cnt, 32));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int scope = 0; scope < 3; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(tab, 0, (size_t)8 * rows * cols * 4)); CK(hipMemset(cnt, 0, 32));
            CK(hipEventRecord(e0));
            if (scope == 0) hipLaunchKernelGGL(add_kernel<0>, dim3(nwg), dim3(256), 0, 0, tab, rows, cols, reps, cnt);
            else if (scope == 1) hipLaunchKernelGGL(add_kernel<1>, dim3(nwg), dim3(256), 0, 0, tab, rows, cols, reps, cnt);
            else hipLaunchKernelGGL(add_kernel<2>, dim3(nwg), dim3(256), 0, 0, tab, rows, cols, reps, cnt);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // check: total of all copies = nwg * reps * cols adds
            static float host[8 * 440 * 512];
            CK(hipMemcpy(host, tab, sizeof(host), hipMemcpyDeviceToHost));
            double tot = 0; for (size_t i = 0; i < sizeof(host) / 4; ++i) tot += host[i];
            unsigned hc[8]; CK(hipMemcpy(hc, cnt, 32, hipMemcpyDeviceToHost));
            const double bytes = (double)nwg * reps * cols * 4;
            printf("%s: %.3f ms  %.2f TB/s of added bytes  sum %.0f (expected %.0f%s)  workgroups per XCD %u %u %u %u %u %u %u %u\n",
                   scope == 0 ? "agent scope    " : (scope == 1 ? "workgroup scope" : "plain stores   "), ms, bytes / ms / 1e9, tot, scope == 2 ? 0.0 : (double)nwg * reps * cols,
                   scope == 2 ? ": stores, no sum" : "", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5], hc[6], hc[7]);
        }
    }
    return 0;
}
