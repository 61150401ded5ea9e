// Micro-benchmark: does per-iteration control code (scalar chains, VALU, branches) overlap with the SIMD
// partner's MFMAs?  24 MFMAs (pair+single) per iteration + NS dependent scalar ops + NV VALU ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NS, int NV, bool BAR>
__global__ __launch_bounds__(512) void kern(float* out, int iters, float a0, float b0, int seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = (float*)smem;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = a0 + i;
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0;
    f32x4 w0 = {b0, b0 + 1, b0 + 2, b0 + 3}, w1 = w0 * 2.f;
    const int lane = threadIdx.x & 63;
    f32x4 a[6];
    unsigned s = __builtin_amdgcn_readfirstlane(seed);
    unsigned v = seed + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NS; ++q) s = __builtin_amdgcn_readfirstlane(s) * 1664525u + 1013904223u;   // scalar chain
#pragma unroll
        for (int q = 0; q < NV; ++q) v = v * 1664525u + 1013904223u;                                    // vector chain
        const int base = (s & 1) * 4096;
        for (int j = 0; j < 6; ++j) a[j] = *(f32x4*)(lds + base + j * 512 + ((lane * 4) ^ (v & 16)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][e], w0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][e], w0[e], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3][e], w1[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4][e], w1[e], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2][e], w0[e], acc2, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[5][e], w1[e], acc2, 0, 0, 0);
        if (BAR) __syncthreads();
    }
    f32x4 r = acc0 + acc1 + acc2;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + (float)(s + v);
}
template <typename K>
void run(const char* name, K k, int threads, float* d) {
    const int blocks = 256 * 4, iters = 4000;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<<<blocks, threads, 100 * 1024>>>(d, 10, 1.f, 2.f, 7);
    hipEventRecord(s);
    k<<<blocks, threads, 100 * 1024>>>(d, iters, 1.f, 2.f, 7);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    double mf = (double)blocks * (threads / 64) * iters * 24.0;
    printf("%-40s %d thr: %.3f ms  %.1f TFLOP/s  %.1f cyc/MFMA/SIMD  (%.0f cyc/iter/wave-pair)\n", name, threads, ms,
           mf * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 * 1024 / mf, ms * 1e-3 * 2.4e9 / (4.0 * iters));
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 512 * 4);
    for (int thr : {512, 256}) {
        run("ctrl 0s 0v bar", kern<0, 0, true>, thr, d);
        run("ctrl 30s 0v bar", kern<30, 0, true>, thr, d);
        run("ctrl 60s 0v bar", kern<60, 0, true>, thr, d);
        run("ctrl 120s 0v bar", kern<120, 0, true>, thr, d);
        run("ctrl 0s 60v bar", kern<0, 60, true>, thr, d);
        run("ctrl 0s 120v bar", kern<0, 120, true>, thr, d);
        run("ctrl 60s 60v bar", kern<60, 60, true>, thr, d);
        run("ctrl 60s 60v nobar", kern<60, 60, false>, thr, d);
    }
    return 0;
}
