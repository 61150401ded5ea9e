// Micro-test for a two-piece fp16 split of fp32 operands on gfx950 ("fewer products"):
// (1) does v_mfma_f32_16x16x32_f16 honour fp16 DENORMAL inputs?  (the second piece of a value below 0.125 is one);
// (2) accuracy of x = x0 + x1 (fp16 pieces), 3 products, on a K = 6912 dot product of scan-like data against float64, next to the
//     fp32 MFMA and the three-piece bf16 split; weights pre-scaled by 2^11.
// hipcc --offload-arch=gfx950 -O3 tools/micro/f16_split.hip -o gpurun_out/f16_split && gpurun_out/f16_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void denorm(float a, float b, float* out) {
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)a; y[i] = (_Float16)b; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
}

// C[16x16] = A[16xK] B[Kx16]; one wave.  mode 0: fp32 MFMA; 2: fp16 two-piece (3 products); 6: bf16 three-piece (6 products)
__global__ void dot(const float* A, const float* B, int K, int mode, float wscale, float* C) {
    const int l = threadIdx.x, li = l & 15, lq = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * K + k + lq], B[(k + lq) * 16 + li], acc, 0, 0, 0);
    } else if (mode == 2) {
        for (int k = 0; k < K; k += 32) {
            f16x8 a[2], b[2];
            for (int i = 0; i < 8; ++i) {
                float x = A[li * K + k + 8 * lq + i], y = B[(k + 8 * lq + i) * 16 + li] * wscale;
                for (int p = 0; p < 2; ++p) {
                    a[p][i] = (_Float16)x; x -= (float)a[p][i];
                    b[p][i] = (_Float16)y; y -= (float)b[p][i];
                }
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) acc[i] /= wscale;
    } else {
        for (int k = 0; k < K; k += 32) {
            bf16x8 a[3], b[3];
            for (int i = 0; i < 8; ++i) {
                float x = A[li * K + k + 8 * lq + i], y = B[(k + 8 * lq + i) * 16 + li];
                for (int p = 0; p < 3; ++p) {
                    a[p][i] = (__bf16)x; x -= (float)a[p][i];
                    b[p][i] = (__bf16)y; y -= (float)b[p][i];
                }
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i) C[(4 * lq + i) * 16 + li] = acc[i];     // (any fixed layout: compared per element below through the same map)
}

int main() {
    float* d;
    hipMalloc(&d, 64 * 4);
    const float tiny[] = {6.1e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f};        // around and below the smallest normal fp16 (6.1e-5)
    for (float t : tiny) {
        denorm<<<1, 64>>>(t, 1.0f, d);
        float h;
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("denormal input %.3g x 1.0, K = 32: MFMA gives %.6g, exact fp16 value x 32 = %.6g\n", t, h, 32.0 * (double)(float)(_Float16)t);
    }
    const int K = 6912;
    std::vector<float> A(16 * K), B(K * 16);
    srand(1);
    auto rn = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
    for (auto& v : A) { double x = rn() * 1.5; v = x > 0 ? (float)x : 0.f; }          // post-ReLU-like activations
    for (auto& v : B) v = (float)(rn() / sqrt(K / 3.0));                                  // kaiming-like weights
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    // reference per (row li, col) in float64; the kernel's output layout is C[(4 lq + i) * 16 + li] = out[col 4 lq + i? ...]: compare
    // through mode 0 instead of decoding the layout: error of each mode = |mode - ref| with ref permuted like mode 0's output
    std::vector<double> ref(256);
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[r * K + k] * (double)B[k * 16 + c];
            ref[r * 16 + c] = s;
        }
    std::vector<float> C0(256), C(256);
    dot<<<1, 64>>>(dA, dB, K, 0, 1.f, dC);
    hipMemcpy(C0.data(), dC, 1024, hipMemcpyDeviceToHost);
    // find the permutation: for each output slot the ref entry closest to mode 0's value
    std::vector<int> perm(256);
    for (int s = 0; s < 256; ++s) { int best = 0; for (int j = 1; j < 256; ++j) if (fabs(ref[j] - C0[s]) < fabs(ref[best] - C0[s])) best = j; perm[s] = best; }
    const struct { int mode; float ws; const char* name; } modes[] = {{0, 1.f, "fp32 MFMA (16x16x4)"}, {6, 1.f, "bf16 x 3 pieces, 6 products"}, {2, 1.f, "fp16 x 2 pieces, 3 products, weights as they are"},
                                                                       {2, 2048.f, "fp16 x 2 pieces, 3 products, weights x 2^11"}};
    for (auto& m : modes) {
        dot<<<1, 64>>>(dA, dB, K, m.mode, m.ws, dC);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double e = 0, mx = 0;
        for (int s = 0; s < 256; ++s) { e = fmax(e, fabs(C[s] - ref[perm[s]])); mx = fmax(mx, fabs(ref[perm[s]])); }
        printf("%-52s max |err| %.3e  (outputs up to %.2f)\n", m.name, e, mx);
    }
    return 0;
}
