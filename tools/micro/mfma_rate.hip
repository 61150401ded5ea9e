// Micro-benchmark: sustained fp32 MFMA rate on gfx950 (16x16x4 and 32x32x2), 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a0, float b0) {
    extern __shared__ char smem[];
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a0, float b0) {
    extern __shared__ char smem[];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) r += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <typename K>
void run(const char* name, K kern, int threads, int nacc, double flop_per_mfma, float* d) {
    const int blocks = 256 * 4, iters = 2000;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    kern<<<blocks, threads, 100 * 1024>>>(d, 10, 1.f, 2.f);
    hipEventRecord(s);
    kern<<<blocks, threads, 100 * 1024>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    double mf = (double)blocks * (threads / 64) * iters * 8.0 * nacc;
    printf("%-28s %d thr/blk nacc=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, threads, nacc, ms,
           mf * flop_per_mfma / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 * 1024 / mf);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 512 * 4);
    run("16x16x4 f32", k16<8>, 512, 8, 2048, d);
    run("16x16x4 f32", k16<8>, 256, 8, 2048, d);
    run("16x16x4 f32", k16<4>, 512, 4, 2048, d);
    run("16x16x4 f32", k16<2>, 512, 2, 2048, d);
    run("16x16x4 f32", k16<1>, 512, 1, 2048, d);
    run("32x32x2 f32", k32<2>, 512, 2, 4096, d);
    run("32x32x2 f32", k32<2>, 256, 2, 4096, d);
    run("32x32x2 f32", k32<1>, 512, 1, 4096, d);
    return 0;
}
