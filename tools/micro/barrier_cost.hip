// Micro-benchmark: cost of s_barrier for a 512-thread workgroup (1 per CU), balanced and unbalanced arrival.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, int iters) {
    extern __shared__ char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1 && wave < 4) __builtin_amdgcn_s_sleep(8);      // half the waves arrive ~512 cycles late
        if (MODE == 2) { if ((i & 1) == (wave >> 2)) __builtin_amdgcn_s_sleep(8); }   // alternate which half is late
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    long long* d; long long h[256];
    hipMalloc(&d, sizeof(h));
    const int iters = 10000;
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        kern<<<256, 512, 140 * 1024>>>(d, iters);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%-40s %.1f cycles per barrier iteration\n", name, s / 256 / iters);
    };
    run(k<0>, "bare barrier loop");
    run(k<1>, "waves 0-3 sleep(8) before each barrier");
    run(k<2>, "alternating half sleeps(8)");
    return 0;
}
