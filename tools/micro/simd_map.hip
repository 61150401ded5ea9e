// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8])
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    extern __shared__ char smem[];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
    unsigned* d; unsigned h[64 * 8];
    hipMalloc(&d, sizeof(h));
    for (int thr : {512, 384}) {
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
        k<<<16, thr, 140 * 1024>>>(d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int b = 0; b < 6; ++b) {
            printf("thr %d block %d: ", thr, b);
            for (int w = 0; w < thr / 64; ++w) printf("w%d:simd%u(cu%u,wv%u) ", w, (h[b * (thr / 64) + w] >> 4) & 3, (h[b * (thr / 64) + w] >> 8) & 15, h[b * (thr / 64) + w] & 15);
            printf("\n");
        }
    }
    return 0;
}
