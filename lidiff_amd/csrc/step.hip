// The boundary between two denoising steps as ONE launch (gfx950): classifier-free guidance, the DPM-Solver++ update and the
// next field's points and voxel coordinates -- DiffCompletion.completion_loop, /root/reference/lidiff/tools/diff_completion_pipeline.py
// :148-153 (guidance), :161-163 (offsets, scheduler.step), :164 + :68-84 (points_to_tensor of the new points).  The reference
// runs this as ~25 elementwise torch launches over [N, 3]; every one of them is a pass over 2-4 MB that cannot fill the chip,
// and they sit on the critical path between two networks.  HBM-bound: 132 B read + 52 B written per point, once.
//
// The arithmetic is the reference's, operation by operation, in the precisions torch-on-the-GPU evaluates it in (no fused
// multiply-adds, a division by a host scalar as the multiplication by its reciprocal that torch's GPU kernels perform):
//   eps    = e_u + w * (e_c - e_u)                                  fp32                                    pipeline:153
//   off    = (double) x_t - x_init                                  fp64 (x_init is float64)  pipeline:162
//   x0     = (off - (double)(sigma_t * eps)) * (1 / alpha_t)        sigma_t * eps in fp32, the rest fp64     epsilon -> data prediction
//   prev   = c_s * off + c_0 * x0 [+ c_1 * ((1 / r0) * (x0 - m1))] + c_z * z        fp64, left to right      sde-dpmsolver++ (App. B)
//   x_new  = x_init + prev                                          fp64                                    pipeline:163
//   feats  = (float) x_new ; coords = (int) rint(feats * (1 / res)) ; batch column likewise                 pipeline:69-72
#include "common.h"

// every product and sum below is rounded on its own, as the reference's separate torch launches round them: no contraction
// into fused multiply-adds anywhere in this file (hipcc's default for device code is -ffp-contract=fast; build.py also compiles
// this file with -ffp-contract=off; tests/test_host.py runs tools/check_asm_regs.py --no-fma over the gfx950 listing)
#pragma clang fp contract(off)

namespace lidiff {

constexpr int kStepBlock = 256;

struct StepCoeffs {
    float w, sigma_t, inv_res;
    double inv_alpha, c_sample, c_m0, c_d1, inv_r0, c_noise;
};

__device__ __forceinline__ int32_t voxel_of(float f, float inv_res) {
    return (int32_t)floorf(rintf(f * inv_res));            // torch.round(x / res) then ME's floor to int32
}

// one point; second: second-order multistep update (m1 = the previous step's data prediction)
__device__ __forceinline__ void step_point(int64_t i, bool second, const float* __restrict__ e_c, const float* __restrict__ e_u,
                                           const float* __restrict__ x_t, const double* __restrict__ x_init,
                                           const double* __restrict__ m1, const double* __restrict__ z, const StepCoeffs& k,
                                           int64_t n_per_batch, int32_t scale_batch, double* __restrict__ x0_out,
                                           float* __restrict__ feats, int32_t* __restrict__ coords) {
    const bool SECOND = second;
    int32_t c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int64_t j = 3 * i + d;
        const float ec = e_c[j], eu = e_u[j];
        const float diff = ec - eu, scaled = k.w * diff, eps = eu + scaled;
        const float se = eps * k.sigma_t;
        const double off = (double)x_t[j] - x_init[j];
        const double x0 = (off - (double)se) * k.inv_alpha;
        const double a = k.c_sample * off, b = k.c_m0 * x0;
        double prev = a + b;
        if (SECOND) {
            const double d1 = k.inv_r0 * (x0 - m1[j]), cterm = k.c_d1 * d1;
            prev = prev + cterm;
        }
        if (z != nullptr) {
            const double zt = k.c_noise * z[j];
            prev = prev + zt;
        }
        const float f = (float)(x_init[j] + prev);
        x0_out[j] = x0;
        feats[j] = f;
        c[d] = voxel_of(f, k.inv_res);
    }
    const int32_t b = (int32_t)(i / n_per_batch);
    const int32_t cb = scale_batch ? voxel_of((float)b, k.inv_res) : b;
    reinterpret_cast<int4*>(coords)[i] = make_int4(cb, c[0], c[1], c[2]);
}

// one thread per point; SECOND: second-order multistep update
template <bool SECOND>
__global__ __launch_bounds__(kStepBlock) void cfg_dpm_step_kernel(const float* __restrict__ e_c, const float* __restrict__ e_u,
                                                                  const float* __restrict__ x_t, const double* __restrict__ x_init,
                                                                  const double* __restrict__ m1, const double* __restrict__ z,
                                                                  StepCoeffs k, int64_t n, int64_t n_per_batch, int32_t scale_batch,
                                                                  double* __restrict__ x0_out, float* __restrict__ feats,
                                                                  int32_t* __restrict__ coords) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    step_point(i, SECOND, e_c, e_u, x_t, x_init, m1, z, k, n_per_batch, scale_batch, x0_out, feats, coords);
}

// DiffCompletion.points_to_tensor (pipeline:68-84) alone: [B, n, 3] points -> float32 features + int32 voxel coordinates
template <typename T>
__global__ __launch_bounds__(kStepBlock) void points_to_field_kernel(const T* __restrict__ pts, float inv_res, int64_t n,
                                                                     int64_t n_per_batch, int32_t scale_batch,
                                                                     float* __restrict__ feats, int32_t* __restrict__ coords) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t c[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float f = (float)pts[3 * i + d];
        feats[3 * i + d] = f;
        c[d] = voxel_of(f, inv_res);
    }
    const int32_t b = (int32_t)(i / n_per_batch);
    const int32_t cb = scale_batch ? voxel_of((float)b, inv_res) : b;
    reinterpret_cast<int4*>(coords)[i] = make_int4(cb, c[0], c[1], c[2]);
}

}  // namespace lidiff

using namespace lidiff;

extern "C" int lidiff_cfg_dpm_step(const float* eps_cond, const float* eps_uncond, float w, const float* x_t, const double* x_init,
                                   const double* m_prev, const double* noise, float sigma_t, double inv_alpha_t, double c_sample,
                                   double c_m0, double c_d1, double inv_r0, double c_noise, float inv_resolution,
                                   int64_t n_points, int64_t n_per_batch, int32_t scale_batch_column, double* x0_out,
                                   float* feats_out, int32_t* coords_out, void* stream) {
    LIDIFF_CHECK_ARG(n_points >= 0 && n_per_batch >= 1, "need n_points >= 0 and n_per_batch >= 1");
    if (n_points == 0) return 0;                 // (an empty field: nothing to do, and its tensors have no storage to point at)
    LIDIFF_CHECK_ARG(eps_cond && eps_uncond && x_t && x_init && x0_out && feats_out && coords_out, "null pointer");
    LIDIFF_CHECK_ARG(((uintptr_t)coords_out & 15) == 0, "coords_out must be 16-byte aligned");
    const StepCoeffs k{w, sigma_t, inv_resolution, inv_alpha_t, c_sample, c_m0, c_d1, inv_r0, c_noise};
    const unsigned grid = (unsigned)ceil_div(n_points, kStepBlock);
    hipStream_t st = (hipStream_t)stream;
    if (m_prev != nullptr)
        cfg_dpm_step_kernel<true><<<grid, kStepBlock, 0, st>>>(eps_cond, eps_uncond, x_t, x_init, m_prev, noise, k, n_points,
                                                               n_per_batch, scale_batch_column, x0_out, feats_out, coords_out);
    else
        cfg_dpm_step_kernel<false><<<grid, kStepBlock, 0, st>>>(eps_cond, eps_uncond, x_t, x_init, nullptr, noise, k, n_points,
                                                                n_per_batch, scale_batch_column, x0_out, feats_out, coords_out);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_points_to_field(const void* points, int32_t is_f64, float inv_resolution, int64_t n_points,
                                      int64_t n_per_batch, int32_t scale_batch_column, float* feats_out, int32_t* coords_out,
                                      void* stream) {
    LIDIFF_CHECK_ARG(n_points >= 0 && n_per_batch >= 1, "need n_points >= 0 and n_per_batch >= 1");
    if (n_points == 0) return 0;
    LIDIFF_CHECK_ARG(points && feats_out && coords_out, "null pointer");
    LIDIFF_CHECK_ARG(((uintptr_t)coords_out & 15) == 0, "coords_out must be 16-byte aligned");
    const unsigned grid = (unsigned)ceil_div(n_points, kStepBlock);
    hipStream_t st = (hipStream_t)stream;
    if (is_f64)
        points_to_field_kernel<double><<<grid, kStepBlock, 0, st>>>((const double*)points, inv_resolution, n_points, n_per_batch,
                                                                    scale_batch_column, feats_out, coords_out);
    else
        points_to_field_kernel<float><<<grid, kStepBlock, 0, st>>>((const float*)points, inv_resolution, n_points, n_per_batch,
                                                                   scale_batch_column, feats_out, coords_out);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}
