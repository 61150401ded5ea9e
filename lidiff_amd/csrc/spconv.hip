// Sparse convolution for gfx950: output-stationary, pair-compacted, fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// One workgroup owns a tile of BM consecutive OUTPUT rows x BN = 16*WN output channels and keeps that
// tile's fp32 accumulators in LDS for the whole kernel-volume loop, so every output row is written to
// HBM exactly once (BatchNorm / residual / ReLU epilogue fused) and there are no global atomics:
// results are deterministic.
//
// For every kernel offset k the tile's column of the neighbour table nbr[k, row0:row0+BM] is compacted
// with a wave ballot into a dense pair list (input row, local output row).  The offset's contribution
//        [n_k pairs x C_in] (gathered rows)  @  W[k] [C_in x BN]
// is issued in 16-pair row blocks, so MFMA work tracks the REAL pair count (< 16 rows of padding per
// offset) instead of BM x 27 as a zero-padded implicit GEMM would.
//
// Decomposition: the WN*WM waves form a WN x WM grid.  Wave (wn, wm) owns output columns
// [16 wn, 16 wn + 16) and the row blocks rb = wm (mod WM) of the stage, so with WM = 1 every wave
// issues exactly the same MFMA stream (no block-dealing imbalance between the SIMDs).
//   * B operand: the wave's [32 k x 16 col] piece of W[k] comes straight from HBM/L2 into 8 VGPRs in
//     MFMA fragment order (weights are pre-packed once by lidiff_spconv_pack_weights: one fully
//     coalesced 2 KB read per wave and stage) and is reused by all of the wave's row blocks.
//   * A operand: the gathered rows of a stage (<= 128 pairs x 32 channels) are written by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass) into a double-buffered
//     [128][32] image whose 16-byte chunks are XOR-swizzled THROUGH THE SOURCE ADDRESS
//     (slot = chunk ^ ((row >> 1) & 7)), which makes the ds_read_b128 fragment reads of all eight
//     waves bank-conflict free.  K is consumed in a permuted order shared by both operands.
//   * A stage = (offset, chunk of <= 128 pairs, 32-channel slab); loads of stage i+1 are issued before
//     the MFMAs of stage i; one barrier per stage.
//   * After the last slab of an offset the register accumulators are added into the LDS tile through
//     the pair list's local output row (each output row occurs at most once per offset and waves own
//     disjoint column / row-block sets, so the LDS read-modify-write is race free).
//
// Replaces ME's ConvolutionForwardGPU (gather -> GEMM -> atomic scatter per offset) behind
// MinkowskiConvolution / MinkowskiConvolutionTranspose; call sites in include/lidiff_amd.h.
#include <type_traits>

#include "common.h"

namespace lidiff {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kSlab = 32;     // input channels per stage
constexpr int kChunk = 128;   // pair rows per stage
constexpr int kAFloats = kChunk * kSlab;   // one A image: 16 KB

struct ConvParams {
    const float* in_a;
    const float* in_b;
    const float* wp;          // packed weights (lidiff_spconv_pack_weights)
    const int32_t* nbr;
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    int64_t m_in, m_out;
    int c_in_a, c_in_b, c_in, c_out, k_vol, relu;
    int tiles_m, tiles_n;
};

template <int BM, int WN, int WM>
struct ConvCfg {
    static constexpr int BN = 16 * WN;
    static constexpr int NW = WN * WM;
    static constexpr int NT = 64 * NW;
    static constexpr int RB = kChunk / 16;       // row blocks per stage
    static constexpr int RBW = RB / WM;          // row blocks per wave
    static constexpr int NCH = BM / kChunk;      // chunks per offset (upper bound)
    static_assert(BM % kChunk == 0 && BM <= 256, "BM");     // out_list is uint8
    static_assert(RB % WM == 0, "WM must divide 8");
    static_assert(NT <= 1024, "workgroup size");

    __host__ __device__ static size_t lds_bytes(int k_vol) {
        size_t b = 2 * (size_t)kAFloats * 4;              // A images (LDS-DMA targets, kept below 64 KB)
        b += (size_t)BM * BN * 4;                         // accumulator tile
        b += (size_t)k_vol * BM * 4;                      // in_list
        b += 32 * 4;                                      // cnt (k_vol <= 27; cnt[31] = #work items)
        b += (size_t)32 * NCH * 4;                        // work list
        b += (size_t)k_vol * BM;                          // out_list (uint8)
        b = (b + 15) & ~(size_t)15;
        return b + 64 * 4;                                // per-lane dummy words for the branch-free flush
    }
};

template <int BM, int WN, int WM, bool VEC>
__global__ __launch_bounds__(64 * WN * WM) void spconv_fwd_kernel(const ConvParams p) {
    using Cfg = ConvCfg<BM, WN, WM>;
    constexpr int BN = Cfg::BN, NT = Cfg::NT, NW = Cfg::NW, RBW = Cfg::RBW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a_buf = reinterpret_cast<float*>(smem);
    float* acc_lds = a_buf + 2 * kAFloats;
    int32_t* in_list = reinterpret_cast<int32_t*>(acc_lds + BM * BN);
    int32_t* cnt = in_list + p.k_vol * BM;
    int32_t* work = cnt + 32;
    uint8_t* out_list = reinterpret_cast<uint8_t*>(work + 32 * Cfg::NCH);
    // float index (relative to acc_lds) of 64 dummy words behind everything else
    const int dummy_off = (int)((Cfg::lds_bytes(p.k_vol) - 64 * 4 - 2 * kAFloats * 4) / 4);

    // XCD-aware tile mapping: the column tiles of one row tile share an XCD (their gathers hit the
    // same L2), consecutive row tiles round-robin over the 8 XCDs.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tm = (g / p.tiles_n) * 8 + xcd;
    if (tm >= p.tiles_m) return;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wn = wave % WN, wm = wave / WN;

    // ---- pair lists: ordered compaction of nbr[k, row0 : row0+rows_here] per offset --------
    if (p.nbr == nullptr) {                      // kernel_size == 1: identity map
        for (int r = tid; r < BM; r += NT) {
            in_list[r] = (int32_t)min(row0 + r, p.m_in - 1);
            out_list[r] = (uint8_t)r;
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        for (int k = wave; k < p.k_vol; k += NW) {
            int pos = 0;
            for (int c = 0; c < BM; c += 64) {
                const int r = c + lane;
                int v = -1;
                if (r < rows_here) v = p.nbr[(int64_t)k * p.m_out + row0 + r];
                const bool valid = v >= 0;
                const unsigned long long m = __ballot(valid);
                if (valid) {
                    const int q = pos + popc_below(m);
                    in_list[k * BM + q] = v;
                    out_list[k * BM + q] = (uint8_t)r;
                }
                pos += __popcll(m);
            }
            if (lane == 0) cnt[k] = pos;
        }
    }
    for (int e = tid; e < BM * BN / 4; e += NT)
        reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // ---- work list: (offset, chunk of <= 128 pairs), ascending k: wave 0, exclusive scan -----
    if (wave == 0) {
        const int c = lane < p.k_vol ? cnt[lane] : 0;
        const int nc = (c + kChunk - 1) / kChunk;
        int incl = nc;
        for (int off = 1; off < 32; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        for (int j = 0; j < nc; ++j)
            work[incl - nc + j] = lane | (j << 8) | (min(kChunk, c - j * kChunk) << 16);
        if (lane == 31) cnt[31] = incl;
    }
    __syncthreads();
    const int nwork = __builtin_amdgcn_readfirstlane(cnt[31]);
    const int nslab = (p.c_in + kSlab - 1) / kSlab;
    const int nit = nwork * nslab;

    // Position in the (work item, K-slab) sequence, kept in SGPRs and advanced incrementally.
    struct Cursor { int wi, slab, k, start, n; };
    auto cursor_load = [&](Cursor& c) {
        if (c.wi < nwork) {
            const int w = __builtin_amdgcn_readfirstlane(work[c.wi]);
            c.k = w & 0xff;
            c.start = ((w >> 8) & 0xff) * kChunk;
            c.n = w >> 16;
        } else {
            c.k = 0; c.start = 0; c.n = 0;
        }
    };
    auto cursor_next = [&](Cursor& c) {
        if (++c.slab == nslab) {
            c.slab = 0;
            ++c.wi;
            cursor_load(c);
        }
    };
    Cursor pf{0, 0, 0, 0, 0}, cur{0, 0, 0, 0, 0};
    cursor_load(pf);
    cur = pf;

    const int nt16 = p.c_out >> 4;
    const float* wp_wave = p.wp + ((size_t)(n0 >> 4) + wn) * 512 + lane * 4;

    // global -> LDS (A, by DMA) and global -> registers (this wave's W fragment) for the stage at `pf`
    auto issue = [&](int buf, f32x4& w0, f32x4& w1) {
        const int k0 = pf.slab * kSlab;
        const float* wsrc = wp_wave + (size_t)(pf.k * nslab + pf.slab) * nt16 * 512;
        w0 = *reinterpret_cast<const f32x4*>(wsrc);
        w1 = *reinterpret_cast<const f32x4*>(wsrc + 256);
        const int32_t* il = in_list + pf.k * BM + pf.start;
        if constexpr (VEC) {
            const bool from_a = k0 < p.c_in_a;                 // uniform: slabs never straddle a|b
            const float* src = from_a ? p.in_a : p.in_b;
            const int cw = from_a ? p.c_in_a : p.c_in_b;
            const int cbase = from_a ? k0 : k0 - p.c_in_a;
            __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(p.m_in * cw * 4), 0x00020000);
            char* dst = reinterpret_cast<char*>(a_buf + buf * kAFloats);
            for (int t = wave; t < kChunk / 8; t += NW) {      // 8 rows x 128 B per wave-instruction
                if (8 * t < pf.n) {
                    const int r = 8 * t + (lane >> 3);
                    const int row = il[min(r, pf.n - 1)];
                    const int ch = (lane & 7) ^ ((r >> 1) & 7);          // source chunk for this LDS slot
                    const bool ok = r < pf.n && cbase + 4 * ch < cw;
                    const int voff = ok ? (row * cw + cbase + 4 * ch) * 4 : (int)0x80000000;   // OOB -> zeros
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + t * 1024), 16, voff, 0, 0, 0);
                }
            }
        } else {
            float* As = a_buf + buf * kAFloats;
            for (int e = tid; e < pf.n * kSlab; e += NT) {
                const int r = e >> 5, c = e & 31, col = k0 + c;
                float v = 0.f;
                if (col < p.c_in) {
                    const int64_t row = il[r];
                    v = (col < p.c_in_a) ? p.in_a[row * p.c_in_a + col] : p.in_b[row * p.c_in_b + (col - p.c_in_a)];
                }
                As[r * kSlab + 4 * ((c >> 2) ^ ((r >> 1) & 7)) + (c & 3)] = v;
            }
        }
        cursor_next(pf);
    };

    f32x4 acc[RBW];
#pragma unroll
    for (int b = 0; b < RBW; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int li = lane & 15, lq = lane >> 4;
    const int aoff = li * kSlab + 4 * (lq ^ ((li >> 1) & 7));      // float offset of chunk lq in row li

    // MFMA half of a stage: the wave's first NB row blocks are active.  MFMA step (j, e) takes
    // k = 16 j + 4 (lane >> 4) + e from both operands; all A fragments are read up front and the
    // MFMAs interleave the NB independent accumulators.
    auto mma = [&](auto nb_tag, const float* As, const f32x4 w0, const f32x4 w1) {
        constexpr int NB = decltype(nb_tag)::value;
        f32x4 a0[NB], a1[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float* q = As + (wm + WM * b) * (16 * kSlab) + aoff;
            a0[b] = *reinterpret_cast<const f32x4*>(q);
            a1[b] = *reinterpret_cast<const f32x4*>(As + (wm + WM * b) * (16 * kSlab) + (aoff ^ 16));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[b][e], w0[e], acc[b], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[b][e], w1[e], acc[b], 0, 0, 0);
    };
    // After the last slab of an offset (chunk): add the NB register blocks into the LDS tile through the
    // pair list's local output row.  Batched and branch-free: NB list words, then 4 NB tile reads, then
    // 4 NB writes (3 LDS round trips); rows beyond n go to a per-lane dummy word.
    auto flush = [&](auto nb_tag) {
        constexpr int NB = decltype(nb_tag)::value;
        const uint8_t* ol = out_list + cur.k * BM + cur.start;
        uint32_t o4[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            o4[b] = *reinterpret_cast<const uint32_t*>(ol + 16 * (wm + WM * b) + 4 * lq);
        int addr[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int prow = 16 * (wm + WM * b) + 4 * lq + r;
                const int orow = (o4[b] >> (8 * r)) & 0xff;
                addr[b][r] = prow < cur.n ? orow * BN + 16 * wn + li : dummy_off + lane;
            }
        float old[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) old[b][r] = acc_lds[addr[b][r]];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_lds[addr[b][r]] = old[b][r] + acc[b][r];
            acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_compute = [&](auto nb_tag, const float* As, const f32x4 w0, const f32x4 w1, bool last) {
        mma(nb_tag, As, w0, w1);
        if (last) flush(nb_tag);
    };
    auto compute_dispatch = [&](int nb, const float* As, const f32x4 w0, const f32x4 w1, bool last) {
        if constexpr (RBW >= 8) {
            if (nb > 4) {
                if (nb == 8) return stage_compute(std::integral_constant<int, 8>{}, As, w0, w1, last);
                if (nb == 7) return stage_compute(std::integral_constant<int, 7>{}, As, w0, w1, last);
                if (nb == 6) return stage_compute(std::integral_constant<int, 6>{}, As, w0, w1, last);
                return stage_compute(std::integral_constant<int, 5>{}, As, w0, w1, last);
            }
        }
        if constexpr (RBW >= 4) {
            if (nb > 2) {
                if (nb == 4) return stage_compute(std::integral_constant<int, 4>{}, As, w0, w1, last);
                return stage_compute(std::integral_constant<int, 3>{}, As, w0, w1, last);
            }
        }
        if constexpr (RBW >= 2) {
            if (nb == 2) return stage_compute(std::integral_constant<int, 2>{}, As, w0, w1, last);
        }
        if (nb == 1) return stage_compute(std::integral_constant<int, 1>{}, As, w0, w1, last);
    };

    f32x4 wc0, wc1, wn0, wn1;
    if (nit > 0) issue(0, wc0, wc1);
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        if (it + 1 < nit) issue((it + 1) & 1, wn0, wn1);
        const int nrb = (cur.n + 15) >> 4;
        const int nb = min(RBW, max(0, (nrb - wm + WM - 1) / WM));
        compute_dispatch(nb, a_buf + (it & 1) * kAFloats, wc0, wc1, cur.slab == nslab - 1);
        cursor_next(cur);
        __syncthreads();                  // stage it+1 landed (vmcnt) and visible; image (it & 1) free
        wc0 = wn0;
        wc1 = wn1;
    }

    // ---- epilogue: BN scale/shift, residual, ReLU; one coalesced float4 store per 4 channels --
    for (int e = tid; e < rows_here * (BN / 4); e += NT) {
        const int r = e / (BN / 4), cq = e % (BN / 4);
        const int col = n0 + 4 * cq;
        float4 v = reinterpret_cast<const float4*>(acc_lds)[r * (BN / 4) + cq];
        if (p.scale) {
            const float4 s = *reinterpret_cast<const float4*>(p.scale + col);
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        }
        if (p.shift) {
            const float4 s = *reinterpret_cast<const float4*>(p.shift + col);
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        const int64_t o = (row0 + r) * p.c_out + col;
        if (p.residual) {
            const float4 s = *reinterpret_cast<const float4*>(p.residual + o);
            v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        }
        if (p.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<float4*>(p.out + o) = v;
    }
}

// W [K, c_in, c_out] row-major  ->  [K][slab][c_out/16][j 0..1][lane 0..63][e 0..3]  with
// k_in = 32 slab + 16 j + 4 (lane >> 4) + e  and  col = 16 nt + (lane & 15); rows beyond c_in are zero.
__global__ void pack_weights_kernel(const float* __restrict__ w, int k_vol, int c_in, int c_out, int nslab,
                                    float* __restrict__ wp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3), lane = (int)((idx >> 2) & 63), j = (int)((idx >> 8) & 1);
    int64_t rest = idx >> 9;
    const int nt16 = c_out >> 4;
    const int nt = (int)(rest % nt16);
    rest /= nt16;
    const int slab = (int)(rest % nslab);
    const int k = (int)(rest / nslab);
    const int kin = 32 * slab + 16 * j + 4 * (lane >> 4) + e;
    const int col = 16 * nt + (lane & 15);
    wp[idx] = kin < c_in ? w[((int64_t)k * c_in + kin) * c_out + col] : 0.f;
}

template <int BM, int WN, int WM, bool VEC>
static int launch_fwd(const ConvParams& p, hipStream_t st) {
    using Cfg = ConvCfg<BM, WN, WM>;
    const size_t lds = Cfg::lds_bytes(p.k_vol);
    LIDIFF_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded");
    auto kern = spconv_fwd_kernel<BM, WN, WM, VEC>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / Cfg::BN;
    const unsigned grid = (unsigned)(ceil_div(q.tiles_m, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

template <int BM, int WN, int WM>
static int dispatch_fwd(const ConvParams& p, bool vec, hipStream_t st) {
    if (vec) return launch_fwd<BM, WN, WM, true>(p, st);
    return launch_fwd<BM, WN, WM, false>(p, st);
}

}  // namespace lidiff

using namespace lidiff;

extern "C" int64_t lidiff_spconv_packed_weight_floats(int32_t k_vol, int32_t c_in, int32_t c_out) {
    return (int64_t)k_vol * ((c_in + kSlab - 1) / kSlab) * kSlab * c_out;
}

extern "C" int lidiff_spconv_pack_weights(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out,
                                          float* w_packed, void* stream) {
    LIDIFF_CHECK_ARG(w != nullptr && w_packed != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && c_in > 0, "kernel volume must be 1..27, c_in > 0");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    const int nslab = (c_in + kSlab - 1) / kSlab;
    const int64_t total = lidiff_spconv_packed_weight_floats(k_vol, c_in, c_out);
    pack_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w, k_vol, c_in, c_out, nslab,
                                                                                         w_packed, total);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_spconv_fwd(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                 const float* w_packed, const int32_t* nbr, int32_t k_vol, int64_t m_in,
                                 int64_t m_out, int32_t c_out, float* out, const float* ep_scale,
                                 const float* ep_shift, const float* residual, int32_t relu,
                                 void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0, "in_a / c_in_a");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(nbr != nullptr || (k_vol == 1 && m_in == m_out), "identity map needs K=1, m_in==m_out");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    LIDIFF_CHECK_ARG(m_out >= 0 && m_in >= 0, "negative rows");
    if (m_out == 0) return 0;
    LIDIFF_CHECK_ARG(m_in > 0, "outputs without inputs");
    ConvParams p{};
    p.in_a = in_a; p.in_b = in_b; p.wp = w_packed; p.nbr = nbr; p.out = out;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.m_in = m_in; p.m_out = m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu;
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(w_packed) && al16(out) && al16(ep_scale) && al16(ep_shift) && al16(residual),
                     "w_packed/out/epilogue pointers must be 16-byte aligned");
    const bool fits32 = m_in * (int64_t)c_in_a * 4 < (1ll << 31) && m_in * (int64_t)c_in_b * 4 < (1ll << 31);
    LIDIFF_CHECK_ARG(fits32, "a feature matrix exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(c_in_b == 0 || c_in_a % kSlab == 0, "with two inputs c_in_a must be a multiple of 32 (slab size)");
    const bool vec = c_in_a % 4 == 0 && c_in_b % 4 == 0 && al16(in_a) && al16(in_b);
    hipStream_t st = (hipStream_t)stream;
    if (c_out % 128 == 0) return dispatch_fwd<128, 8, 1>(p, vec, st);
    if (c_out % 96 == 0) return dispatch_fwd<128, 6, 1>(p, vec, st);
    if (c_out % 64 == 0) return dispatch_fwd<128, 4, 2>(p, vec, st);
    if (c_out % 32 == 0) return dispatch_fwd<128, 2, 4>(p, vec, st);
    return dispatch_fwd<128, 1, 8>(p, vec, st);
}

extern "C" int lidiff_spconv_bwd_w(const float*, int32_t, const float*, int32_t, const float*,
                                   const int32_t*, int32_t, int64_t, int64_t, int32_t, float*, void*) {
    set_error("lidiff_spconv_bwd_w: not built yet");
    return 3;
}
