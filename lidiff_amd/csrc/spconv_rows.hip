// Convolutions whose every output row has exactly ONE pair, as streaming row GEMMs for gfx950:
//   (1) identity maps -- kernel_size 1, and the centre offset of the centre + tail scheme:
//           out[r, :] = epilogue( [in_a | in_b][r, :] @ W  (+ the row's tail rows) )
//   (2) pair lists grouped by kernel offset (lidiff_spconv_fwd_pairs) -- the transposed kernel_size-2 / stride-2 maps of the
//       decoder (each fine voxel has one parent) and the tail pass of centre + tail (one output row per pair):
//           out[pair_out[p], :] = epilogue( [in_a | in_b][pair_in[p], :] @ W[offset of p] )
// Nothing to compact, one offset per row block -- which the tile kernel of spconv.hip runs through its whole pair-list
// machinery (zeroed LDS accumulator tile, pair lists, a barrier per channel slab, a flush, an epilogue that reads the tile
// back): 295 us for 96 -> 96 on the 357 000 rows of the bench scan's stride-1 level, where the rows take 55 us to stream at
// HBM speed and 42 us to multiply.
//
// Here a workgroup of 8 waves keeps the [C_in x 16 NT16] column tile of W (of its offset) in LDS (fragment order, straight
// copy of the packed weights: <= 96 KB) and every wave walks over 16-row blocks on its own:
//   * the block's rows come straight from HBM into MFMA operand registers -- lane (li, lq) loads the 16 bytes
//     [row li][16 g + 4 lq .. + 3] of every 16-channel group g, which is exactly the permuted K order the packed
//     weights use (k = 16 g + 4 (lane >> 4) + e), no LDS, no barrier;
//   * v_mfma_f32_16x16x4_f32 with the operands SWAPPED (W fragment as the A operand, the rows as B), i.e. the
//     transposed product: lane (li, lq) then holds out[row li][16 nt + 4 lq .. + 3] -- four consecutive channels,
//     one 16-byte store, and the epilogue's scale / shift / residual / tail rows are 16-byte loads of the same shape;
//     the sum over k has the same terms in the same order as the tile kernel's (same instruction, same (g, e)
//     sequence), so the two kernels agree to the last bit;
//   * the rows of the next block (and, for pair lists, the row indices of the block after it) are requested before the
//     current one is multiplied.
// No barrier after the weights have landed.  For identity maps the classifier-free-guidance replicas are simply more rows
// (stacked matrices are contiguous; only the tail CSR is indexed per replica); pair lists take them as a grid dimension.
#include "spconv.h"

namespace lidiff {

// (kernels and helpers live in lidiff:: itself, not in an anonymous namespace: profiler tools cut kernel names at the first
// parenthesis, and "lidiff::(anonymous namespace)::..." would lose its name)
constexpr int kRowsWaves = 8;

struct PairList {
    const int32_t* pair_in;     // input row of every pair, pairs grouped by offset
    const int32_t* pair_out;    // its output row (nullable: the pair's own position)
    const int32_t* offset_ptr;  // [k_vol + 1] pair ranges of the offsets
    int64_t n_pairs;            // rows of pair_in / pair_out: offset ranges are clamped to it (a no-op unless the list is a
                                //   bounded tail map that overflowed -- the caller then redoes the step, nothing is overrun)
};

template <int NJ, int NT16, bool GATHER>
__global__ __launch_bounds__(64 * kRowsWaves) void spconv_rows_kernel(const ConvParams p_launch, const PairList pl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ConvParams p = p_launch;
    float* w_s = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nt16 = p.c_out >> 4, tiles_n = nt16 / NT16;
    const int nt0 = (GATHER ? blockIdx.y % tiles_n : blockIdx.y) * NT16;
    constexpr int NS = (NJ + 1) / 2;                       // 32-channel slabs of the packed weights
    int64_t seg_lo = 0, seg_hi = p.m_out * p.replicas;     // positions of this workgroup's offset
    if constexpr (!GATHER) {
        // row count on the device (m_out: bound and replica pitch): the replicas' valid rows are not contiguous any more, each
        // replica is a grid slice of its own
        if (p.d_m_out) {
            seg_lo = (int64_t)blockIdx.z * p.m_out;
            seg_hi = seg_lo + valid_rows(p);
        }
    }
    if constexpr (GATHER) {
        const int k = blockIdx.z, rep = blockIdx.y / tiles_n;
        seg_lo = min((int64_t)pl.offset_ptr[k], pl.n_pairs);
        seg_hi = min((int64_t)pl.offset_ptr[k + 1], pl.n_pairs);
        p.wp += (int64_t)k * NS * nt16 * 512;
        p.in_a += (int64_t)rep * p.m_in * p.c_in_a;
        if (p.in_b) p.in_b += (int64_t)rep * p.m_in * p.c_in_b;
        p.out += (int64_t)rep * p.m_out * p.c_out;
        if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    }
    const int64_t nblk = (seg_hi - seg_lo + 15) >> 4;
    const int64_t gw = (int64_t)gridDim.x * kRowsWaves;
    if ((int64_t)blockIdx.x * kRowsWaves >= nblk) return;  // (uniform) nothing of this offset left for this workgroup
    // W column tile -> LDS as [slab][nt][j][lane][e]: per slab one contiguous run of the packed array
    for (int s = 0; s < NS; ++s) {
        const float4* src = reinterpret_cast<const float4*>(p.wp + ((int64_t)s * nt16 + nt0) * 512);
        float4* dst = reinterpret_cast<float4*>(w_s + s * NT16 * 512);
        for (int e = tid; e < NT16 * 128; e += 64 * kRowsWaves) dst[e] = src[e];
    }
    // BatchNorm scale / shift of the tile's columns behind it (an LDS read per use instead of a vector-memory load:
    // with fp32 MFMAs queued every VMEM instruction costs the SIMD ~80 issue cycles)
    float* ss_s = w_s + NS * NT16 * 512;
    if (tid < NT16 * 16) {
        ss_s[tid] = p.scale ? p.scale[nt0 * 16 + tid] : 1.f;
        ss_s[NT16 * 16 + tid] = p.shift ? p.shift[nt0 * 16 + tid] : 0.f;
    }
    const int nja = p.c_in_a >> 4;
    int64_t b = (int64_t)blockIdx.x * kRowsWaves + wave;

    // (input row, output row) of this lane's position in block blk; positions behind the segment repeat its last one
    struct Rows { int64_t in, out; };
    auto block_rows = [&](int64_t blk) {
        const int64_t pos = min(seg_lo + blk * 16 + li, seg_hi - 1);
        if constexpr (GATHER) return Rows{(int64_t)pl.pair_in[pos], pl.pair_out ? (int64_t)pl.pair_out[pos] : pos};
        else return Rows{pos, pos};
    };
    auto load_block = [&](int64_t row, f32x4* a) {
        const float* ra = p.in_a + row * p.c_in_a + 4 * lq;
        const float* rb = p.in_b ? p.in_b + row * p.c_in_b + 4 * lq : ra;
#pragma unroll
        for (int g = 0; g < NJ; ++g)
            a[g] = *reinterpret_cast<const f32x4*>(g < nja ? ra + 16 * g : rb + 16 * (g - nja));
    };
    // the next block's rows are requested before the current block is multiplied where the registers allow it
    // (4 waves per SIMD stay resident: <= 128 VGPRs); the wide shapes rely on the other resident waves instead
    constexpr bool PF = NJ + NT16 <= 12;
    f32x4 a[NJ], an[PF ? NJ : 1];
    Rows cur = block_rows(min(b, nblk - 1));
    Rows nxt = block_rows(min(b + gw, nblk - 1));
    if (b < nblk) load_block(cur.in, a);
    __syncthreads();                                       // W tile in LDS; the only barrier
    for (; b < nblk; b += gw) {
        const bool more = b + gw < nblk;
        if constexpr (PF) {
            if (more) load_block(nxt.in, an);
        }
        // the indices of the block after the next: one iteration ahead of the row loads that depend on them
        const Rows nn = block_rows(min(b + 2 * gw, nblk - 1));
        f32x4 acc[NT16];
#pragma unroll
        for (int nt = 0; nt < NT16; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < NJ; ++g) {
            f32x4 w[NT16];
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt)
                w[nt] = *reinterpret_cast<const f32x4*>(w_s + ((g >> 1) * NT16 + nt) * 512 + (g & 1) * 256 + lane * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NT16; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt][e], a[g][e], acc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);             // keep the W fragments of ONE group live, not of all of them
        }
        // ---- epilogue: lane (li, lq) owns out[row][16 nt + 4 lq .. + 3] ----
        if (seg_lo + b * 16 + li < seg_hi) {
            const int64_t row = cur.out;
            const int col0 = nt0 * 16 + 4 * lq;
            if constexpr (!GATHER) {
                if (p.tail) {                              // the non-centre offsets' contributions, fixed order
                    const int rep = (int)(row / p.m_out);
                    const int64_t rloc = row - (int64_t)rep * p.m_out;
                    const float* tb = p.tail + (int64_t)rep * p.tail_rows * p.c_out + col0;
                    for (int q = p.tail_ptr[rloc], qe = min(p.tail_ptr[rloc + 1], (int)p.tail_rows); q < qe; ++q) {
                        const float* tr = tb + (int64_t)p.tail_idx[q] * p.c_out;
#pragma unroll
                        for (int nt = 0; nt < NT16; ++nt) acc[nt] += *reinterpret_cast<const f32x4*>(tr + 16 * nt);
                    }
                }
            }
            const int64_t o = row * p.c_out + col0;
#pragma unroll
            for (int nt = 0; nt < NT16; ++nt) {
                f32x4 v = acc[nt];
                if (p.scale) v *= *reinterpret_cast<const f32x4*>(ss_s + 4 * lq + 16 * nt);
                if (p.shift) v += *reinterpret_cast<const f32x4*>(ss_s + NT16 * 16 + 4 * lq + 16 * nt);
                if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + o + 16 * nt);
                if (p.relu) {
                    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                }
                *reinterpret_cast<f32x4*>(p.out + o + 16 * nt) = v;
            }
        }
        if (more) {
            if constexpr (PF) {
#pragma unroll
                for (int g = 0; g < NJ; ++g) a[g] = an[g];
            } else {
                load_block(nxt.in, a);
            }
        }
        cur = nxt;
        nxt = nn;
    }
}

template <int NJ, int NT16, bool GATHER>
static int launch_rows(const ConvParams& p, const PairList& pl, int64_t n_pos, hipStream_t st) {
    constexpr int NS = (NJ + 1) / 2;
    const size_t lds = (size_t)NS * NT16 * 512 * 4 + (size_t)NT16 * 32 * 4;     // W tile + scale / shift
    auto kern = spconv_rows_kernel<NJ, NT16, GATHER>;
    static thread_local bool configured = false;
    if (!configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const int tiles_n = p.c_out / (16 * NT16);
    const int per_cu = (int)min((size_t)2, (size_t)(160 * 1024) / lds);
    const int64_t nblk = ceil_div(n_pos, 16);
    if (!GATHER) {
        // every wave the same number of 16-row blocks (+-1), with as many workgroups as the chip holds at once
        // (row count on the device: one grid slice per replica, each sized for the bound)
        const int reps_z = p.d_m_out ? p.replicas : 1;
        const int64_t nblk_z = p.d_m_out ? ceil_div(p.m_out, 16) : nblk;
        const int64_t slots = max((int64_t)1, (int64_t)256 * per_cu / (tiles_n * reps_z)) * kRowsWaves;
        const int64_t per_wave = ceil_div(nblk_z, slots);
        const unsigned gx = (unsigned)ceil_div(nblk_z, per_wave * kRowsWaves);
        hipLaunchKernelGGL(kern, dim3(gx, (unsigned)tiles_n, (unsigned)reps_z), dim3(64 * kRowsWaves), lds, st, p, pl);
    } else {
        // the pair counts of the offsets live on the device: every offset gets the workgroups an even split would need
        // (x 2: a transposed stride-2 map is even, a tail map is not); a workgroup strides over its offset's blocks and
        // leaves at once when there are none left for it
        const int64_t slots = max((int64_t)1, (int64_t)256 * per_cu / (tiles_n * p.replicas));
        const int64_t per_k = max((int64_t)1, min(ceil_div(2 * slots, p.k_vol), ceil_div(nblk, kRowsWaves)));
        hipLaunchKernelGGL(kern, dim3((unsigned)per_k, (unsigned)(tiles_n * p.replicas), (unsigned)p.k_vol),
                           dim3(64 * kRowsWaves), lds, st, p, pl);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

template <int NJ, bool GATHER>
static int dispatch_rows(const ConvParams& p, const PairList& pl, int64_t n_pos, hipStream_t st) {
    if (p.c_out % 128 == 0) return launch_rows<NJ, 8, GATHER>(p, pl, n_pos, st);
    if (p.c_out % 96 == 0) return launch_rows<NJ, 6, GATHER>(p, pl, n_pos, st);
    if (p.c_out % 64 == 0) return launch_rows<NJ, 4, GATHER>(p, pl, n_pos, st);
    return launch_rows<NJ, 2, GATHER>(p, pl, n_pos, st);
}

template <bool GATHER>
static int dispatch_rows_nj(const ConvParams& p, const PairList& pl, int64_t n_pos, hipStream_t st) {
    switch (p.c_in >> 4) {
        case 2: return dispatch_rows<2, GATHER>(p, pl, n_pos, st);
        case 4: return dispatch_rows<4, GATHER>(p, pl, n_pos, st);
        case 6: return dispatch_rows<6, GATHER>(p, pl, n_pos, st);
        case 8: return dispatch_rows<8, GATHER>(p, pl, n_pos, st);
        case 12:
            if constexpr (!GATHER) return dispatch_rows<12, false>(p, pl, n_pos, st);
            break;
    }
    set_error("spconv_rows: unsupported channel count");
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Thin inputs (C_in <= 4: the networks' stems see the 3 point coordinates, minkunet.py:94): 2 x 27 x 3 x 32 FLOPs per output row
// are nothing -- the layer is the 27 table entries and the 128 output bytes per row.  The MFMA kernels cannot even vectorise a
// 12-byte row (scalar gather path: 105 us for the stem's first convolution on the bench scan).  Plain VALU kernel: one lane per
// output row with its 32 accumulators, table reads coalesced over the rows, W (K x C_in x 32 floats) broadcast from LDS, the
// wave's 64 x 32 output tile turned through LDS into 1 KB stores.  Offsets ascending, channels ascending, fused multiply-adds.
constexpr int kThinCo = 32;

__global__ __launch_bounds__(256) void spconv_thin_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w_s = reinterpret_cast<float*>(smem);                       // [K][c_in][32]
    float* o_s = w_s + p.k_vol * p.c_in * kThinCo;                     // [4 waves][64 rows][33]
    // out of the packed layout (one 32-channel slab, two 16-column tiles): W[k][ci][col] sits at lane = col & 15, e = ci
    for (int e = threadIdx.x; e < p.k_vol * p.c_in * kThinCo; e += 256) {
        const int col = e % kThinCo, ci = (e / kThinCo) % p.c_in, k = e / (kThinCo * p.c_in);
        w_s[e] = p.wp[((int64_t)k * 2 + (col >> 4)) * 512 + (col & 15) * 4 + ci];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = p.m_out * p.replicas;
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < rows;
    const int rep = live ? (int)(row / p.m_out) : 0;
    const int64_t o = live ? row - (int64_t)rep * p.m_out : 0;
    const float* in = p.in_a + (int64_t)rep * p.m_in * p.c_in;
    float acc[kThinCo];
#pragma unroll
    for (int c = 0; c < kThinCo; ++c) acc[c] = 0.f;
    for (int k = 0; k < p.k_vol; ++k) {
        const int idx = !live ? -1 : (p.nbr ? p.nbr[(int64_t)k * p.m_out + o] : (int)o);
        if (idx < 0) continue;
        const float* x = in + (int64_t)idx * p.c_in;
        const float* wk = w_s + k * p.c_in * kThinCo;
        for (int ci = 0; ci < p.c_in; ++ci) {
            const float xv = x[ci];
#pragma unroll
            for (int c = 0; c < kThinCo; ++c) acc[c] = fmaf(xv, wk[ci * kThinCo + c], acc[c]);
        }
    }
    float* tile = o_s + wave * 64 * 33;
#pragma unroll
    for (int c = 0; c < kThinCo; ++c) tile[lane * 33 + c] = acc[c];
    // the exchange is between lanes of ONE wave: the hardware executes a wave's LDS accesses in order, but the compiler may only
    // be told so -- a wavefront-scope release fence plus a wave barrier pins the ds_writes in front of the ds_reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t row0 = (int64_t)blockIdx.x * 256 + wave * 64;
    for (int e = lane; e < 64 * kThinCo; e += 64) {                   // 64 consecutive floats of the wave's 64 x 32 tile per pass
        const int r = e >> 5, c = e & 31;
        if (row0 + r >= rows) break;
        float v = tile[r * 33 + c];
        if (p.scale) v *= p.scale[c];
        if (p.shift) v += p.shift[c];
        const int64_t oo = (row0 + r) * kThinCo + c;
        if (p.residual) v += p.residual[oo];
        if (p.relu) v = fmaxf(v, 0.f);
        p.out[oo] = v;
    }
}

static bool rows_shapes_ok(int c_in_a, int c_in_b, int c_out, bool gather) {
    if (c_in_a % 16 != 0 || c_in_b % 16 != 0 || c_out % 32 != 0) return false;
    const int nj = (c_in_a + c_in_b) >> 4;
    return nj == 2 || nj == 4 || nj == 6 || nj == 8 || (nj == 12 && !gather);
    // (256 input channels -- 64-column W tiles -- were built and measured: 348 vs 340 us on the 256 -> 256 transposed layer,
    // 260 vs 254 on 256 -> 128: the tile kernel keeps them)
}

bool rows_kernel_applies(const ConvParams& p) {
    if (p.nbr != nullptr || p.row_order != nullptr || p.k_vol != 1 || p.m_in != p.m_out) return false;
    return rows_shapes_ok(p.c_in_a, p.c_in_b, p.c_out, false);
}

bool thin_kernel_applies(const ConvParams& p) {
    return p.in_b == nullptr && p.c_in_b == 0 && p.c_in_a >= 1 && p.c_in_a <= 4 && p.c_out == kThinCo && p.row_order == nullptr &&
           p.tail == nullptr && (p.nbr != nullptr || (p.k_vol == 1 && p.m_in == p.m_out));
}

int launch_fwd_thin(const ConvParams& p, hipStream_t st) {
    const size_t lds = ((size_t)p.k_vol * p.c_in * kThinCo + 4 * 64 * 33) * sizeof(float);
    const int64_t rows = p.m_out * p.replicas;
    hipLaunchKernelGGL(spconv_thin_kernel, dim3((unsigned)ceil_div(rows, 256)), dim3(256), lds, st, p);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int launch_fwd_rows(const ConvParams& p, hipStream_t st) {
    return dispatch_rows_nj<false>(p, PairList{nullptr, nullptr, nullptr, 0}, p.m_out * p.replicas, st);
}

}  // namespace lidiff

using namespace lidiff;

extern "C" int32_t lidiff_spconv_fwd_pairs_supported(int32_t c_in_a, int32_t c_in_b, int32_t c_out) {
    return rows_shapes_ok(c_in_a, c_in_b, c_out, true) ? 1 : 0;
}

extern "C" int lidiff_spconv_fwd_pairs(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                       const float* w_packed, int32_t k_vol, const int32_t* pair_in,
                                       const int32_t* pair_out, const int32_t* offset_ptr, int64_t n_pairs, int64_t m_in,
                                       int64_t m_out, int32_t c_out, float* out, const float* ep_scale,
                                       const float* ep_shift, const float* residual, int32_t relu, int32_t replicas,
                                       void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0 && w_packed != nullptr && out != nullptr, "null pointer");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(pair_in != nullptr && offset_ptr != nullptr, "pair_in / offset_ptr");
    LIDIFF_CHECK_ARG(replicas >= 1 && n_pairs >= 0 && m_in >= 0 && m_out >= 0, "negative size");
    LIDIFF_CHECK_ARG(rows_shapes_ok(c_in_a, c_in_b, c_out, true),
                     "channel widths: multiples of 16 with c_in in {32, 64, 96, 128}, c_out a multiple of 32");
    if (n_pairs == 0 || m_out == 0) return 0;
    LIDIFF_CHECK_ARG(m_in > 0, "pairs without inputs");
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(in_a) && al16(in_b) && al16(w_packed) && al16(out) && al16(residual),
                     "feature / weight pointers must be 16-byte aligned");
    ConvParams p{};
    p.in_a = in_a; p.in_b = in_b; p.wp = w_packed; p.out = out;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.m_in = m_in; p.m_out = m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu; p.replicas = replicas;
    return dispatch_rows_nj<true>(p, PairList{pair_in, pair_out, offset_ptr, n_pairs}, n_pairs, (hipStream_t)stream);
}
