// The network's last lines as ONE launch: SparseTensor.slice(field).F followed by the head MLP
//     Linear(C, H) -> LeakyReLU(slope) -> Linear(H, O)
// (minkunet.py:390 `self.last`, :497 `self.last(y4.slice(x).F)`; H = 20, O = 3 in LiDiff).  torch runs it as a row gather that
// writes [points, C] (138 MB for 2 x 180 000 points x 96 channels), two GEMMs with N = 20 and N = 3 -- which the library serves at
// 0.4 TB/s of its input -- and an activation in between: 0.9 ms per denoising step on the main stream.  Here a thread owns one POINT:
// it reads its voxel's feature row straight from the voxel matrix (16-byte loads; the rows of a scan's 180 000 points are ~150 000
// distinct voxel rows, L2-resident), keeps the H hidden sums in registers -- the weights are wave-uniform and arrive as scalar loads
// from the [C][H] transposed matrix -- and writes O floats.  HBM-bound by construction: C x 4 bytes read + O x 4 written per point.
// Every sum runs over the channels in ascending order with fused multiply-adds: deterministic, and independent of the number of
// rows (the same bits whether the voxel matrix is handed over at its exact size or at its bound).
#include "common.h"

namespace lidiff {

template <int H, int O>
__global__ __launch_bounds__(256) void slice_head_kernel(const float* __restrict__ feats, const int64_t* __restrict__ inv,
                                                         int64_t n_per, int64_t m_per, int replicas, int c,
                                                         const float* __restrict__ w1t, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2, float slope,
                                                         float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_per * replicas) return;
    const int64_t rep = i / n_per;                                        // replica r reads voxel rows [r m_per, (r + 1) m_per)
    const float4* row = reinterpret_cast<const float4*>(feats + (inv[i - rep * n_per] + rep * m_per) * c);
    float acc[H];
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = b1[j];
    auto chunk = [&](const float4 x, int c4) {
        const float* w = w1t + (size_t)4 * c4 * H;                         // wave-uniform: scalar loads
        const float xe[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < H; ++j) acc[j] = fmaf(xe[e], w[e * H + j], acc[j]);
    };
    const int nc4 = c / 4;
    int c4 = 0;
    for (; c4 + 4 <= nc4; c4 += 4) {                                       // four 16-byte loads of the row in flight
        const float4 x0 = row[c4], x1 = row[c4 + 1], x2 = row[c4 + 2], x3 = row[c4 + 3];
        chunk(x0, c4); chunk(x1, c4 + 1); chunk(x2, c4 + 2); chunk(x3, c4 + 3);
    }
    for (; c4 < nc4; ++c4) chunk(row[c4], c4);
#pragma unroll
    for (int j = 0; j < H; ++j) acc[j] = acc[j] > 0.f ? acc[j] : acc[j] * slope;
#pragma unroll
    for (int o = 0; o < O; ++o) {
        float s = b2[o];
#pragma unroll
        for (int j = 0; j < H; ++j) s = fmaf(acc[j], w2[o * H + j], s);
        out[i * O + o] = s;
    }
}

}  // namespace lidiff

using namespace lidiff;

extern "C" int32_t lidiff_slice_head_supported(int32_t c, int32_t hidden, int32_t c_out) {
    return c > 0 && c % 4 == 0 && hidden == 20 && c_out == 3;
}

extern "C" int lidiff_slice_head(const float* feats, const int64_t* inverse, int64_t n_points, int64_t m_rows, int32_t replicas,
                                 int32_t c, const float* w1_t, const float* b1, int32_t hidden, const float* w2, const float* b2,
                                 int32_t c_out, float slope, float* out, void* stream) {
    LIDIFF_CHECK_ARG(lidiff_slice_head_supported(c, hidden, c_out), "supported: c % 4 == 0, hidden == 20, c_out == 3");
    LIDIFF_CHECK_ARG(n_points >= 0 && m_rows >= 0 && replicas >= 1, "sizes");
    LIDIFF_CHECK_ARG(((uintptr_t)feats & 15) == 0, "feats must be 16-byte aligned");
    if (n_points == 0) return 0;
    LIDIFF_CHECK_ARG(feats && inverse && w1_t && b1 && w2 && b2 && out, "null pointer");
    const int64_t total = n_points * replicas;
    slice_head_kernel<20, 3><<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(
        feats, inverse, n_points, m_rows, replicas, c, w1_t, b1, w2, b2, slope, out);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}
