// Does LDS-DMA (buffer_load_dwordx4 ... lds, M0 base) reach LDS addresses above 64 KB on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* as3_p;
__global__ void k(const float* in, float* out, int n, int lds_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) ((float*)smem)[i] = -1.f;
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (as3_p)(smem + lds_off + w * 1024), 16, threadIdx.x * 16, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = ((float*)(smem + lds_off))[i];
}
int main() {
    float *din, *dout, h[1024], hin[1024];
    for (int i = 0; i < 1024; ++i) hin[i] = (float)i;
    hipMalloc(&din, 4096); hipMalloc(&dout, 4096);
    hipMemcpy(din, hin, 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int off : {0, 32768, 61440, 65536, 98304, 131072, 155648}) {
        k<<<1, 256, 160 * 1024>>>(din, dout, 1024, off);
        hipMemcpy(h, dout, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += h[i] != (float)i;
        printf("LDS-DMA to offset %6d: %s (%d wrong of 1024, first values %g %g)\n", off, bad ? "WRONG" : "ok", bad, h[0], h[1]);
    }
    return 0;
}
