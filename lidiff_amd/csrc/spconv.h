// Shared declarations of the sparse-convolution kernels (spconv.hip: tile kernels for every shape; spconv_rows.hip: one pair per
// output row; spconv_bf16.hip: bf16 operands; spconv_split3.hip: fp32 results from three-way split bf16 operands).
#pragma once
#include <type_traits>

#include "common.h"

namespace lidiff {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kChunk = 128;   // pair rows per stage

struct ConvParams {
    const float* in_a;
    const float* in_b;
    const float* wp;          // packed weights (lidiff_spconv_pack_weights)
    const int32_t* nbr;
    const int32_t* row_order;   // nullable: tile rows -> output rows
    float* out;
    void* out_planes;           // nullable (spconv_split3.hip): the output cut into three bf16 pieces, [m_out][3][c_out]
    const float* scale;
    const float* shift;
    const float* residual;
    const float* tail;          // nullable: rows added to the sum BEFORE the epilogue, through a CSR over output rows:
    const int32_t* tail_ptr;    //   out[o] += sum of tail[tail_idx[q]] for q in [tail_ptr[o], tail_ptr[o+1])
    const int32_t* tail_idx;
    int64_t tail_rows;          //   rows of `tail` per replica
    int64_t m_in, m_out;
    const int32_t* d_m_out;     // nullable: the number of VALID output rows lives on the device (<= m_out, which stays the row
                                //   pitch of nbr / of the stacked replicas and bounds the grid): tiles behind it leave at once
    int c_in_a, c_in_b, c_in, c_out, k_vol, relu;
    int tiles_m, tiles_n, flags, replicas;
    float out_scale;          // spconv_fwd_split3_kernel<BN, 2>: the inverse of the power of two the fp16 weights were packed with
    int32_t* status;          //   ... and where a value beyond fp16's range is reported (LIDIFF_STATUS_F16_RANGE; nullable)
    int probe;                // LIDIFF_CONV_PROBE builds only: bit 0 = no A gather, 1 = no W loads,
                              // 2 = no barrier, 4 = no flush
    long long* timeline;      // LIDIFF_CONV_PROBE builds only: 8 cycle counters per workgroup (tools/conv_probe.py)
};

#ifdef LIDIFF_CONV_PROBE
#define PROBE(bit) (p.probe & (bit))
#define STAMP(var) const long long var = __builtin_readcyclecounter()
#else
#define PROBE(bit) false
#define STAMP(var)
#endif


template <int I>
using ic = std::integral_constant<int, I>;

// valid output rows of a launch: the device-side count when the caller's row count is only a bound (host-read-free steps)
__device__ __forceinline__ int64_t valid_rows(const ConvParams& p) {
    return p.d_m_out ? min((int64_t)*p.d_m_out, p.m_out) : p.m_out;
}

// weight gradient (spconv.hip, spconv_bf16.hip): pairs per chunk, pair slices per offset, the slice-ordered reduction
constexpr int kDwPairs = 64;
int64_t dw_slices(int c_in, int c_out, int k_vol, int64_t n_pairs, int cit, int cot);
// The slices x k_vol workgroups of a launch (per tile of dW) are SLOTS handed to the offsets in proportion to their pair counts:
// a slot is `per` pairs of one offset (whole chunks), offset k owns ceil(P_k / per) consecutive slots.  per is sized so that all
// offsets together need at most slices x k_vol slots (the surplus ones stay idle).
int64_t dw_pairs_per_slot(int64_t n_pairs, int64_t slices, int k_vol);
// which offset a slot belongs to, and which of the offset's slots it is (false: an idle slot)
__device__ __forceinline__ bool dw_slot_offset(const int32_t* __restrict__ offset_ptr, int k_vol, int64_t m_ident, int64_t per,
                                               int slot, int& k, int& local, int64_t& p_lo, int64_t& p_hi) {
    int cum = 0;
    for (int kk = 0; kk < k_vol; ++kk) {
        const int64_t lo = offset_ptr ? offset_ptr[kk] : 0, hi = offset_ptr ? offset_ptr[kk + 1] : m_ident;
        const int sk = (int)((hi - lo + per - 1) / per);
        if (slot < cum + sk) { k = kk; local = slot - cum; p_lo = lo; p_hi = hi; return true; }
        cum += sk;
    }
    return false;
}
__global__ void dw_reduce_kernel(const float* __restrict__ part, int64_t n_k, int k_vol, const int32_t* __restrict__ offset_ptr,
                                 int64_t m_ident, int64_t per, float* __restrict__ dw);

// spconv_rows.hip: identity maps (kernel_size 1 / the centre pass) as a streaming row GEMM.
bool rows_kernel_applies(const ConvParams& p);
int launch_fwd_rows(const ConvParams& p, hipStream_t st);
// ... and inputs of <= 4 channels (the stems) as a plain VALU kernel.
bool thin_kernel_applies(const ConvParams& p);
int launch_fwd_thin(const ConvParams& p, hipStream_t st);

}  // namespace lidiff
