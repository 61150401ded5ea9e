// Micro-benchmark: do VALU instructions of the SIMD partner wave overlap with fp32 (and bf16) MFMAs?
// Waves 0-3 of a 512-thread workgroup issue MFMAs back to back; waves 4-7 (their SIMD partners) issue NV
// independent VALU ops (v_add_u32 / v_fma_f32) per MFMA-wave iteration.  Reports the MFMA waves' cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV>   // KIND 0: f32 16x16x4 MFMA, 1: bf16 16x16x32 MFMA, 2: no MFMA (VALU only)
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters) {
    extern __shared__ char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 2.f;
    bf16x8 ah = {1, 2, 3, 4, 5, 6, 7, 8}, bh = {8, 7, 6, 5, 4, 3, 2, 1};
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (KIND != 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
                    }
            }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + (float)v[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}
template <typename K>
void run(const char* name, K kern, long long* d, float* sink) {
    const int iters = 2000;
    long long h[256 * 8];
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    kern<<<256, 512, 140 * 1024>>>(d, sink, iters);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, vv = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < 4; ++w) m += h[b * 8 + w]; for (int w = 4; w < 8; ++w) vv += h[b * 8 + w]; }
    printf("%-46s MFMA waves %.1f cyc/iter (64 MFMAs)   VALU waves %.1f cyc/iter\n", name, m / 1024 / iters, vv / 1024 / iters);
}
int main() {
    long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 4);
    run("f32 MFMA, partner idle", k<0, 0>, d, sink);
    run("f32 MFMA, partner 64 v_add / iter", k<0, 64>, d, sink);
    run("f32 MFMA, partner 256 v_add / iter", k<0, 256>, d, sink);
    run("f32 MFMA, partner 512 v_add / iter", k<0, 512>, d, sink);
    run("no MFMA, 256 v_add / iter", k<2, 256>, d, sink);
    run("no MFMA, 512 v_add / iter", k<2, 512>, d, sink);
    run("bf16 MFMA, partner idle", k<1, 0>, d, sink);
    run("bf16 MFMA, partner 256 v_add / iter", k<1, 256>, d, sink);
    run("bf16 MFMA, partner 512 v_add / iter", k<1, 512>, d, sink);
    return 0;
}
