// Micro-benchmark for the "bf16-split operands" lever of DESIGN.md 4: (1) sustained rate of v_mfma_f32_16x16x32_bf16
// against v_mfma_f32_16x16x4_f32 on gfx950, per fp32-equivalent product when one product costs 3 (two-way split,
// hi*hi + hi*lo + lo*hi) or 6 (three-way split) bf16 MFMAs; (2) the accuracy of those splits on a K = 6912 dot
// product (27 offsets x 256 channels) of N(0,1) data, against a float64 reference.
// hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_split.hip -o gpurun_out/bf16_split && gpurun_out/bf16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, bool BF>
__global__ __launch_bounds__(512) void rate(float* out, int iters, float a0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a0 + threadIdx.x + i); bb[i] = (__bf16)(a0 * i); }
    const float af = a0 + threadIdx.x, bf = a0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = BF ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// C[16x16] = A[16xK] B[Kx16]; one wave.  mode 0: fp32 MFMA; 1: bf16; 3: two-way split (3 MFMAs); 6: three-way (6)
__global__ void dot(const float* A, const float* B, int K, int mode, float* C) {
    const int l = threadIdx.x, li = l & 15, lq = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * K + k + lq], B[(k + lq) * 16 + li], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                float x = A[li * K + k + 8 * lq + i], y = B[(k + 8 * lq + i) * 16 + li];
                for (int p = 0; p < 3; ++p) {                       // successive bf16 pieces of the value
                    a[p][i] = (__bf16)x; x -= (float)a[p][i];
                    b[p][i] = (__bf16)y; y -= (float)b[p][i];
                }
            }
            // smallest terms first
            if (mode == 6) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
            }
            if (mode >= 3) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 4; ++r) C[(4 * lq + r) * 16 + li] = acc[r];
}

template <typename K>
double run_rate(const char* name, K kern, double flop, float* d) {
    const int blocks = 1024, iters = 2000;
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    kern<<<blocks, 512>>>(d, 10, 1.f);
    hipEventRecord(s);
    kern<<<blocks, 512>>>(d, iters, 1.f);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double mf = (double)blocks * 8 * iters * 8.0 * 8;
    const double tf = mf * flop / (ms * 1e-3) / 1e12;
    printf("%-30s %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", name, ms, tf, ms * 1e-3 * 2.4e9 * 1024 / mf);
    return tf;
}

int main() {
    float* d;
    hipMalloc(&d, 1024 * 512 * 4);
    const double f32 = run_rate("16x16x4 f32", rate<8, false>, 2048, d);
    const double b16 = run_rate("16x16x32 bf16", rate<8, true>, 16384, d);
    printf("fp32-equivalent rate: two-way split (3 MFMAs) %.1f TFLOP/s = %.2fx fp32; three-way (6 MFMAs) %.1f = %.2fx\n",
           b16 / 3, b16 / 3 / f32, b16 / 6, b16 / 6 / f32);
    const int K = 6912;
    std::vector<float> A(16 * K), B(K * 16);
    srand(1);
    auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
    for (auto& x : A) x = (float)nrm();
    for (auto& x : B) x = (float)(0.05 * nrm());
    std::vector<double> ref(256, 0.0);
    double scale = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 16 + j]; sa += fabs((double)A[i * K + k] * B[k * 16 + j]); }
            ref[i * 16 + j] = s;
            scale += sa / 256;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("K = %d dot products, errors relative to sum |a_k b_k| (= %.3f):\n", K, scale);
    for (int mode : {0, 1, 3, 6}) {
        dot<<<1, 64>>>(dA, dB, K, mode, dC);
        std::vector<float> C(256);
        hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
        double mx = 0, mean = 0;
        for (int i = 0; i < 256; ++i) { const double e = fabs(C[i] - ref[i]) / scale; mx = fmax(mx, e); mean += e / 256; }
        printf("  %-34s max %.3e  mean %.3e\n", mode == 0 ? "fp32 MFMA" : mode == 1 ? "bf16 MFMA" : mode == 3 ? "bf16 two-way split (3 MFMAs)" : "bf16 three-way split (6 MFMAs)", mx, mean);
    }
    return 0;
}
