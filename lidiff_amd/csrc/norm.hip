// Training-mode BatchNorm over the rows of a feature matrix [M, C] (MinkowskiBatchNorm = nn.BatchNorm1d on F: minkunet.py:23,
// 59, 79; train.py under models.py:180-217) for gfx950.  HBM-bound: every kernel is one pass over [M, C].
//
// torch's channels-last batch-norm kernels reach 0.4-0.5 TB/s on these shapes (M = 10^5..4 10^5 rows of 32..256 channels:
// batch_norm_collect_statistics 9 ms and batch_norm_backward_reduce 12.5 ms of a 154 ms bf16 training step).  Here a
// workgroup owns a slab of rows; a thread owns 4 consecutive channels (one 16-byte load per row) and walks down the slab
// with DOUBLE accumulators -- the statistics are exact to fp64 rounding whatever the magnitude of the mean, and the
// per-workgroup partial sums are combined in a fixed order by a second kernel: deterministic.
//   forward :  stats  (sum x, sum x^2 per channel -> mean, biased var)          reads X
//              apply  y = (x - mean) * invstd * gamma + beta  [optional ReLU]    reads X, writes Y
//   backward:  reduce (sum dy, sum dy (x - mean) per channel)                    reads dY, X  [dY masked by y > 0 with ReLU]
//              apply  dx = (dy - sum_dy / M - (x - mean) invstd^2 sum_dy_xmu / M) invstd gamma     reads dY, X, writes dX
#include "common.h"

namespace lidiff {

constexpr int kBnBlock = 256;
constexpr int kBnMaxBlocks = 2048;

__host__ __device__ inline int bn_rows_per_block(int c) { return kBnBlock / (c / 4); }

// grid: nblk workgroups; workgroup b owns rows [b * per, (b + 1) * per) ; thread = (row lane, channel quad)
template <bool BWD, bool RELU>
__global__ __launch_bounds__(kBnBlock) void bn_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ y, const float* __restrict__ mean,
                                                            int64_t m, int c, int64_t per, double* __restrict__ part) {
    extern __shared__ double sm[];                       // [rows per block][c] x 2
    const int cq = c >> 2, rl = threadIdx.x / cq, q = threadIdx.x % cq, rpb = kBnBlock / cq;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (rl < rpb) {
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BWD) mu = reinterpret_cast<const float4*>(mean)[q];
        const int64_t lo = (int64_t)blockIdx.x * per, hi = min(m, lo + per);
        for (int64_t r = lo + rl; r < hi; r += rpb) {
            const float4 xv = reinterpret_cast<const float4*>(x + r * c)[q];
            if (!BWD) {
                a[0] += xv.x; a[1] += xv.y; a[2] += xv.z; a[3] += xv.w;
                b[0] += (double)xv.x * xv.x; b[1] += (double)xv.y * xv.y; b[2] += (double)xv.z * xv.z; b[3] += (double)xv.w * xv.w;
            } else {
                float4 g = reinterpret_cast<const float4*>(dy + r * c)[q];
                if (RELU) {
                    const float4 yv = reinterpret_cast<const float4*>(y + r * c)[q];
                    g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
                }
                a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
                b[0] += (double)g.x * (xv.x - mu.x); b[1] += (double)g.y * (xv.y - mu.y);
                b[2] += (double)g.z * (xv.z - mu.z); b[3] += (double)g.w * (xv.w - mu.w);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sm[(rl * c + 4 * q + e) * 2] = a[e];
            sm[(rl * c + 4 * q + e) * 2 + 1] = b[e];
        }
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += kBnBlock) {            // row lanes in order: fixed summation order
        double sa = 0, sb = 0;
        for (int r = 0; r < rpb; ++r) { sa += sm[(r * c + ch) * 2]; sb += sm[(r * c + ch) * 2 + 1]; }
        part[((int64_t)ch * gridDim.x + blockIdx.x) * 2] = sa;          // [channel][workgroup]: a channel's partials are contiguous
        part[((int64_t)ch * gridDim.x + blockIdx.x) * 2 + 1] = sb;
    }
}

// one wave per channel: lane l adds the partials of workgroups l, l + 64, ... in order, then a fixed butterfly over the lanes
__device__ __forceinline__ void bn_channel_sums(const double* __restrict__ part, int nblk, int ch, double& s, double& s2) {
    s = 0; s2 = 0;
    for (int b = threadIdx.x; b < nblk; b += kWave) {
        s += part[((int64_t)ch * nblk + b) * 2];
        s2 += part[((int64_t)ch * nblk + b) * 2 + 1];
    }
    for (int off = kWave / 2; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        s2 += __shfl_down(s2, off);
    }
}

// running estimates as nn.BatchNorm1d keeps them: r = (1 - momentum) r + momentum x, the variance unbiased (x count / (count - 1));
// the two roundings of torch's mul_ / add_ pair
__device__ __forceinline__ void bn_update_running(float* __restrict__ running_mean, float* __restrict__ running_var, int ch,
                                                  float mean, float var, double count, float momentum) {
#pragma clang fp contract(off)
    if (running_mean != nullptr) running_mean[ch] = running_mean[ch] * (1.0f - momentum) + mean * momentum;
    if (running_var != nullptr) {
        const float unbias = (float)(count / (count - 1.0)) * momentum;
        running_var[ch] = running_var[ch] * (1.0f - momentum) + var * unbias;
    }
}

__global__ __launch_bounds__(kWave) void bn_finish_stats_kernel(const double* __restrict__ part, int nblk, int c, int64_t m,
                                                                 float eps, float* __restrict__ mean, float* __restrict__ var,
                                                                 float* __restrict__ invstd, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, float momentum) {
    const int ch = blockIdx.x;
    double s, s2;
    bn_channel_sums(part, nblk, ch, s, s2);
    if (threadIdx.x != 0) return;
    const double mu = s / (double)m;
    double v = s2 / (double)m - mu * mu;
    if (v < 0) v = 0;
    mean[ch] = (float)mu;
    var[ch] = (float)v;                                  // biased (the normaliser); the running estimate takes the unbiased one
    invstd[ch] = (float)(1.0 / sqrt(v + (double)eps));
    bn_update_running(running_mean, running_var, ch, (float)mu, (float)v, (double)m, momentum);
}

__global__ __launch_bounds__(kWave) void bn_finish_bwd_kernel(const double* __restrict__ part, int nblk, int c,
                                                               float* __restrict__ sum_dy, float* __restrict__ sum_dy_xmu) {
    const int ch = blockIdx.x;
    double s, s2;
    bn_channel_sums(part, nblk, ch, s, s2);
    if (threadIdx.x != 0) return;
    sum_dy[ch] = (float)s;
    sum_dy_xmu[ch] = (float)s2;
}

// SyncBatchNorm (train.py:90): the per-channel raw sums leave the device-side reduction as fp64 [2][c] so that the host side can
// all-reduce them over the process group (together with the row count) before anything is derived from them
__global__ __launch_bounds__(kWave) void bn_finish_sums_kernel(const double* __restrict__ part, int nblk, int c,
                                                                double* __restrict__ sums, double count) {
    const int ch = blockIdx.x;
    double s, s2;
    bn_channel_sums(part, nblk, ch, s, s2);
    if (threadIdx.x != 0) return;
    sums[ch] = s;
    sums[c + ch] = s2;
    if (ch == 0 && count >= 0) sums[2 * c] = count;      // this rank's row count travels behind the sums (forward only)
}

__global__ void bn_stats_from_sums_kernel(const double* __restrict__ sums, int c, float eps, float* __restrict__ mean,
                                          float* __restrict__ var, float* __restrict__ invstd, float* __restrict__ running_mean,
                                          float* __restrict__ running_var, float momentum) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double m = sums[2 * c];                        // the (all-reduced) row count travels behind the sums
    const double mu = sums[ch] / m;
    double v = sums[c + ch] / m - mu * mu;
    if (v < 0) v = 0;
    mean[ch] = (float)mu;
    var[ch] = (float)v;
    invstd[ch] = (float)(1.0 / sqrt(v + (double)eps));
    bn_update_running(running_mean, running_var, ch, (float)mu, (float)v, m, momentum);
}

__global__ void bn_bwd_from_sums_kernel(const double* __restrict__ sums, int c, float* __restrict__ sum_dy,
                                        float* __restrict__ sum_dy_xmu) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    sum_dy[ch] = (float)sums[ch];
    sum_dy_xmu[ch] = (float)sums[c + ch];
}

// four floats -> four bf16 (round to nearest even), as lidiff_cast_bf16 rounds
__device__ __forceinline__ uint2 pack_bf16x4(const float4 v) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t f = {v.x, v.y, v.z, v.w};
    return __builtin_bit_cast(uint2, __builtin_convertvector(f, bf16x4_t));
}

template <bool RELU>
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ residual, int64_t total4, int cq, float* __restrict__ y,
                                uint2* __restrict__ y16 = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int q = (int)(i % cq);
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
    float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gamma) g = reinterpret_cast<const float4*>(gamma)[q];
    if (beta) bb = reinterpret_cast<const float4*>(beta)[q];
    float4 o;
    o.x = (xv.x - mu.x) * is.x * g.x + bb.x; o.y = (xv.y - mu.y) * is.y * g.y + bb.y;
    o.z = (xv.z - mu.z) * is.z * g.z + bb.z; o.w = (xv.w - mu.w) * is.w * g.w + bb.w;
    if (residual) {
        const float4 r = reinterpret_cast<const float4*>(residual)[i];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    reinterpret_cast<float4*>(y)[i] = o;
    if (y16) y16[i] = pack_bf16x4(o);                     // the bf16 shadow the next convolution gathers (bf16 training)
}

template <bool RELU>
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                                    const float* __restrict__ sum_dy_xmu, int64_t total4, int cq, float inv_m,
                                    float* __restrict__ dx, float* __restrict__ d_residual,
                                    const double* __restrict__ d_count = nullptr, uint2* __restrict__ dx16 = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    if (d_count) inv_m = (float)(1.0 / *d_count);         // SyncBatchNorm: the all-reduced row count stays on the device
    const int q = (int)(i % cq);
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    if (RELU) {
        const float4 yv = reinterpret_cast<const float4*>(y)[i];
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    if (d_residual) reinterpret_cast<float4*>(d_residual)[i] = g;          // the residual branch's gradient: dy behind the ReLU
    if (dx == nullptr) return;
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[q], is = reinterpret_cast<const float4*>(invstd)[q];
    const float4 sd = reinterpret_cast<const float4*>(sum_dy)[q], sx = reinterpret_cast<const float4*>(sum_dy_xmu)[q];
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (gamma) w = reinterpret_cast<const float4*>(gamma)[q];
    auto one = [&](float gv, float xx, float m_, float i_, float sd_, float sx_, float w_) {
        const float k = sx_ * i_ * i_ * inv_m;                 // projection on (x - mean)
        return (gv - sd_ * inv_m - (xx - m_) * k) * i_ * w_;
    };
    float4 o;
    o.x = one(g.x, xv.x, mu.x, is.x, sd.x, sx.x, w.x); o.y = one(g.y, xv.y, mu.y, is.y, sd.y, sx.y, w.y);
    o.z = one(g.z, xv.z, mu.z, is.z, sd.z, sx.z, w.z); o.w = one(g.w, xv.w, mu.w, is.w, sd.w, sx.w, w.w);
    reinterpret_cast<float4*>(dx)[i] = o;
    if (dx16) dx16[i] = pack_bf16x4(o);                   // ... and of the gradient the convolution in front of this layer gathers
}

static int bn_blocks(int64_t m, int c, int64_t* per) {
    const int rpb = bn_rows_per_block(c);
    int64_t nblk = ceil_div(m, (int64_t)rpb * 8);            // >= 8 rows per thread
    if (nblk > kBnMaxBlocks) nblk = kBnMaxBlocks;
    if (nblk < 1) nblk = 1;
    *per = ceil_div(m, nblk);
    return (int)ceil_div(m, *per);
}

}  // namespace lidiff

using namespace lidiff;

extern "C" int64_t lidiff_bn_workspace_bytes(int32_t c) { return (int64_t)kBnMaxBlocks * c * 2 * (int64_t)sizeof(double); }

static bool bn_shape_ok(int64_t m, int c) { return m >= 1 && c >= 4 && c % 4 == 0 && c <= 1024 && kBnBlock / (c / 4) >= 1; }

extern "C" int lidiff_bn_stats(const float* x, int64_t m, int32_t c, float eps, float* mean, float* var, float* invstd,
                               float* running_mean, float* running_var, float momentum, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(x && mean && var && invstd && workspace, "null pointer");
    LIDIFF_CHECK_ARG(bn_shape_ok(m, c), "need m >= 1 and c a multiple of 4 in [4, 1024]");
    LIDIFF_CHECK_ARG(((uintptr_t)x & 15) == 0, "x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int64_t per;
    const int nblk = bn_blocks(m, c, &per);
    const size_t lds = (size_t)bn_rows_per_block(c) * c * 2 * sizeof(double);
    bn_reduce_kernel<false, false><<<nblk, kBnBlock, lds, st>>>(x, nullptr, nullptr, nullptr, m, c, per, (double*)workspace);
    bn_finish_stats_kernel<<<(unsigned)c, kWave, 0, st>>>((const double*)workspace, nblk, c, m, eps, mean, var, invstd, running_mean,
                                                          running_var, momentum);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

// -- SyncBatchNorm pieces: local sums -> [all-reduce by the caller] -> statistics -------------------------------------------------
// (The synchronised pieces take m == 0 as well: a rank that holds no row of some layer still takes part in the layer's
//  collective -- with zero sums and a row count of zero -- as torch's SyncBatchNorm lets it)
extern "C" int lidiff_bn_sums(const float* x, int64_t m, int32_t c, double* sums, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(sums && workspace && (x || m == 0), "null pointer");
    LIDIFF_CHECK_ARG(m >= 0 && bn_shape_ok(m > 0 ? m : 1, c), "need m >= 0 and c a multiple of 4 in [4, 1024]");
    LIDIFF_CHECK_ARG(((uintptr_t)x & 15) == 0, "x must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        LIDIFF_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)(2 * c + 1) * sizeof(double), st));
        return 0;
    }
    int64_t per;
    const int nblk = bn_blocks(m, c, &per);
    const size_t lds = (size_t)bn_rows_per_block(c) * c * 2 * sizeof(double);
    bn_reduce_kernel<false, false><<<nblk, kBnBlock, lds, st>>>(x, nullptr, nullptr, nullptr, m, c, per, (double*)workspace);
    bn_finish_sums_kernel<<<(unsigned)c, kWave, 0, st>>>((const double*)workspace, nblk, c, sums, (double)m);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_bn_stats_from_sums(const double* sums, int32_t c, float eps, float* mean, float* var, float* invstd,
                                         float* running_mean, float* running_var, float momentum, void* stream) {
    LIDIFF_CHECK_ARG(sums && mean && var && invstd, "null pointer");
    LIDIFF_CHECK_ARG(c >= 1, "need c >= 1");
    bn_stats_from_sums_kernel<<<(unsigned)ceil_div(c, 256), 256, 0, (hipStream_t)stream>>>(sums, c, eps, mean, var, invstd, running_mean,
                                                                                           running_var, momentum);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_bn_bwd_sums(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                                  double* sums, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(mean && sums && workspace && ((dy && x) || m == 0), "null pointer");
    LIDIFF_CHECK_ARG(m >= 0 && bn_shape_ok(m > 0 ? m : 1, c), "need m >= 0 and c a multiple of 4 in [4, 1024]");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) {
        LIDIFF_CHECK_HIP(hipMemsetAsync(sums, 0, (size_t)(2 * c) * sizeof(double), st));
        return 0;
    }
    int64_t per;
    const int nblk = bn_blocks(m, c, &per);
    const size_t lds = (size_t)bn_rows_per_block(c) * c * 2 * sizeof(double);
    if (y_relu) bn_reduce_kernel<true, true><<<nblk, kBnBlock, lds, st>>>(x, dy, y_relu, mean, m, c, per, (double*)workspace);
    else bn_reduce_kernel<true, false><<<nblk, kBnBlock, lds, st>>>(x, dy, nullptr, mean, m, c, per, (double*)workspace);
    bn_finish_sums_kernel<<<(unsigned)c, kWave, 0, st>>>((const double*)workspace, nblk, c, sums, -1.0);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_bn_bwd_apply(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                                   const float* invstd, const float* gamma, const double* sums, const double* count,
                                   float* sum_dy, float* sum_dy_xmu, float* dx, float* d_residual, void* dx_bf16, void* stream) {
    LIDIFF_CHECK_ARG(mean && invstd && sums && count && sum_dy && sum_dy_xmu && ((dy && x) || m == 0), "null pointer");
    LIDIFF_CHECK_ARG(m >= 0 && bn_shape_ok(m > 0 ? m : 1, c), "need m >= 0 and c a multiple of 4 in [4, 1024]");
    hipStream_t st = (hipStream_t)stream;
    bn_bwd_from_sums_kernel<<<(unsigned)ceil_div(c, 256), 256, 0, st>>>(sums, c, sum_dy, sum_dy_xmu);
    if (m > 0 && (dx != nullptr || d_residual != nullptr)) {
        const int64_t total4 = m * (c / 4);
        const unsigned grid = (unsigned)ceil_div(total4, 256);
        if (y_relu) bn_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(dy, x, y_relu, mean, invstd, gamma, sum_dy, sum_dy_xmu, total4, c / 4, 0.f, dx, d_residual, count, (uint2*)dx_bf16);
        else bn_bwd_apply_kernel<false><<<grid, 256, 0, st>>>(dy, x, nullptr, mean, invstd, gamma, sum_dy, sum_dy_xmu, total4, c / 4, 0.f, dx, d_residual, count, (uint2*)dx_bf16);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_bn_apply(const float* x, int64_t m, int32_t c, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, const float* residual, int32_t relu, float* y,
                               void* y_bf16, void* stream) {
    LIDIFF_CHECK_ARG(mean && invstd && ((x && y) || m == 0), "null pointer");
    LIDIFF_CHECK_ARG(m >= 0 && bn_shape_ok(m > 0 ? m : 1, c), "need m >= 0 and c a multiple of 4 in [4, 1024]");
    if (m == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total4 = m * (c / 4);
    const unsigned grid = (unsigned)ceil_div(total4, 256);
    if (relu) bn_apply_kernel<true><<<grid, 256, 0, st>>>(x, mean, invstd, gamma, beta, residual, total4, c / 4, y, (uint2*)y_bf16);
    else bn_apply_kernel<false><<<grid, 256, 0, st>>>(x, mean, invstd, gamma, beta, residual, total4, c / 4, y, (uint2*)y_bf16);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_bn_bwd(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                             const float* invstd, const float* gamma, float* sum_dy, float* sum_dy_xmu, float* dx,
                             float* d_residual, void* workspace, void* dx_bf16, void* stream) {
    LIDIFF_CHECK_ARG(dy && x && mean && invstd && sum_dy && sum_dy_xmu && workspace, "null pointer");
    LIDIFF_CHECK_ARG(bn_shape_ok(m, c), "need m >= 1 and c a multiple of 4 in [4, 1024]");
    hipStream_t st = (hipStream_t)stream;
    int64_t per;
    const int nblk = bn_blocks(m, c, &per);
    const size_t lds = (size_t)bn_rows_per_block(c) * c * 2 * sizeof(double);
    if (y_relu) bn_reduce_kernel<true, true><<<nblk, kBnBlock, lds, st>>>(x, dy, y_relu, mean, m, c, per, (double*)workspace);
    else bn_reduce_kernel<true, false><<<nblk, kBnBlock, lds, st>>>(x, dy, nullptr, mean, m, c, per, (double*)workspace);
    bn_finish_bwd_kernel<<<(unsigned)c, kWave, 0, st>>>((const double*)workspace, nblk, c, sum_dy, sum_dy_xmu);
    if (dx != nullptr || d_residual != nullptr) {
        const int64_t total4 = m * (c / 4);
        const unsigned grid = (unsigned)ceil_div(total4, 256);
        const float inv_m = 1.0f / (float)m;
        if (y_relu) bn_bwd_apply_kernel<true><<<grid, 256, 0, st>>>(dy, x, y_relu, mean, invstd, gamma, sum_dy, sum_dy_xmu, total4, c / 4, inv_m, dx, d_residual, nullptr, (uint2*)dx_bf16);
        else bn_bwd_apply_kernel<false><<<grid, 256, 0, st>>>(dy, x, nullptr, mean, invstd, gamma, sum_dy, sum_dy_xmu, total4, c / 4, inv_m, dx, d_residual, nullptr, (uint2*)dx_bf16);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}
