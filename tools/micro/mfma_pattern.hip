// Micro-benchmark: the conv kernel's MFMA issue patterns (pair / single chains, A fragments from LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: pair(0,1)+single(2) from registers; 1: same, A fragments re-read from LDS every iteration;
// 2: 3 accumulators round-robin (no dependent neighbours); 3: mode 1 + a barrier per iteration
template <int MODE>
__global__ __launch_bounds__(512) void kern(float* out, int iters, float a0, float b0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds = (float*)smem;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = a0 + i;
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0;
    f32x4 w0 = {b0, b0 + 1, b0 + 2, b0 + 3}, w1 = w0 * 2.f;
    const int lane = threadIdx.x & 63;
    f32x4 a[6];
    for (int j = 0; j < 6; ++j) a[j] = *(f32x4*)(lds + j * 512 + lane * 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) {
            const int base = (it & 1) * 4096;
            for (int j = 0; j < 6; ++j) a[j] = *(f32x4*)(lds + base + j * 512 + ((lane * 4) ^ (it & 16)));
        }
        if (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][e], w0[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][e], w0[e], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2][e], w0[e], acc2, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3][e], w1[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4][e], w1[e], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[5][e], w1[e], acc2, 0, 0, 0);
            }
        } else {
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
        }
        if (MODE == 3) __syncthreads();
    }
    f32x4 r = acc0 + acc1 + acc2;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] + r[1] + r[2] + r[3];
}
template <typename K>
void run(const char* name, K k, int threads, float* d) {
    const int blocks = 256 * 4, iters = 4000;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    k<<<blocks, threads, 100 * 1024>>>(d, 10, 1.f, 2.f);
    hipEventRecord(s);
    k<<<blocks, threads, 100 * 1024>>>(d, iters, 1.f, 2.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    double mf = (double)blocks * (threads / 64) * iters * 24.0;
    printf("%-44s %d thr: %.3f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD\n", name, threads, ms, mf * 2048 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 * 1024 / mf);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 512 * 4);
    for (int thr : {512, 256}) {
        run("pair+single, regs", kern<0>, thr, d);
        run("pair+single, A from LDS each iter", kern<1>, thr, d);
        run("3 accs round-robin, regs", kern<2>, thr, d);
        run("pair+single, LDS + barrier each iter", kern<3>, thr, d);
    }
    return 0;
}
