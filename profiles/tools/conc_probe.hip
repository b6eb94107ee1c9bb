#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin_kernel(long long ticks, int* out) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4);
    const long long ticks = 100000;  // 1 ms at 100 MHz
    for (int grid : {32, 192, 256, 1024}) {
        for (int threads : {256, 512}) {
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s1, ticks, d); hipDeviceSynchronize();
            double t = now();
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s1, ticks, d);
            hipDeviceSynchronize();
            double one = now() - t;
            t = now();
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s1, ticks, d);
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s2, ticks, d);
            hipDeviceSynchronize();
            double two = now() - t;
            t = now();
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s1, ticks, d);
            hipLaunchKernelGGL(spin_kernel, dim3(grid), dim3(threads), 0, s1, ticks, d);
            hipDeviceSynchronize();
            double same = now() - t;
            printf("grid %4d x %3d: one %.3f ms, two streams %.3f ms, same stream twice %.3f ms\n", grid, threads, one, two, same);
        }
    }
    return 0;
}
