// Sparse convolution, fp32 in / fp32 out, with the contraction carried out on the bf16 matrix pipe from THREE-WAY SPLIT operands
// (round 6; MinkowskiConvolution of the dense levels -- minkunet.py:53-66,184-259 -- in the eval-mode fused plan).
//
// Why: on gfx950 the fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 rate) and shares the SIMD's issue with
// every other instruction (tools/micro/mfma_valu_share.hip), so the native kernel of spconv.hip cannot pass ~0.63 of that peak
// (matrix pipe 71 % busy + 22 % vector-memory issue + 6 % VALU: profiles/r05_pmc_mfma.txt; ping-pong schedules lose).
// An fp32 value is the exact sum of three bf16 pieces (8 + 8 + 8 significand bits, round to nearest even at every cut):
//        x = x0 + x1 + x2,   w = w0 + w1 + w2,
//        x w = x0 w0 + (x0 w1 + x1 w0) + (x0 w2 + x1 w1 + x2 w0) + [terms below 2^-24 |x w|: dropped],
// six bf16 x bf16 products, each EXACT in the MFMA's fp32 accumulator -- measured error of a K = 6 912 dot product relative to
// sum |x w|: 1.27e-7 against 1.13e-7 for v_mfma_f32_16x16x4_f32 itself (profiles/r01_bf16_split_micro.txt): the result is an fp32
// convolution to fp32 accuracy, not a bf16 one.  Six v_mfma_f32_16x16x32_bf16 do the work of sixteen v_mfma_f32_16x16x4_f32
// in 6/16 of the matrix-pipe time, and -- unlike fp32 MFMAs -- leave the SIMD's other issue slots alone.
//
// Layout: the pieces are cut ONCE by the producer -- the feature matrix a dense convolution reads exists as bf16 [M][3][C]
// (lidiff_split3_rows, or the `out_planes` epilogue of this kernel), the weights as three packed planes
// (lidiff_spconv_pack_weights_bf16, planes = 3) -- so the kernel moves 6 bytes per operand element and converts nothing.
//
// Kernel: the wide register-tile decomposition of spconv_bf16.hip (spconv_fwd_bf16_wide_kernel): a workgroup owns 256 output rows
// x 128 columns, multiplies ALL its rows for every offset that occurs in the tile (a missing neighbour is gathered as zeros: an
// out-of-range request moves no bytes), so every product lands in a fixed accumulator REGISTER -- no accumulator tile in LDS, no
// pair lists, no flush.  A stage = (offset, 32 input channels): 3 x 16 KB of rows + 24 KB of W fragments by LDS-DMA,
// double-buffered (144 KB), one barrier per stage; a wave (2 row groups x 4 column groups) multiplies 128 rows x 32 columns:
// 8 x 2 blocks x 6 products = 96 MFMAs per stage.  Two levels of fp32 sums (per offset, then over the offsets: as the native kernel).
// A per-tile, per-offset mask of the 16-row blocks that hold a neighbour skips the others (requests and MFMAs alike); the caller
// sorts the map's rows by their neighbour sets (lidiff_row_mask_keys) so that such blocks are common.  The kernel executes at 0.9 of
// a dense hipBLASLt bf16 GEMM on the same chip.
//
// NP = 2 (opt-in, never the default): the same kernel on TWO fp16 pieces per operand and three products -- 22-bit operands, half the
// matrix work, fp16's range (reported through LIDIFF_STATUS_F16_RANGE); see lidiff_amd.h and profiles/r06_f16x2.txt.
#include "spconv.h"

namespace lidiff {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int kStatusF16Range = 8;                          // LIDIFF_STATUS_F16_RANGE (include/lidiff_amd.h)

__device__ __forceinline__ f32x4 s3_mfma(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 s3_mfma(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void s3_dma16(const i32x4 rsrc, const unsigned lds_addr, const int voff, const int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// a wave-uniform value into a scalar register, whatever register class the compiler computed it in (the "s" operands of the
// request above: `__builtin_amdgcn_readfirstlane` of a value the compiler has PROVED uniform is folded away, and the operand
// then arrives in a vector register -- "invalid operand for instruction")
__device__ __forceinline__ int to_sgpr(const int v) {
    int s;
    asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(v));
    return s;
}

// x -> (bf16(x), bf16(x - p0), bf16(x - p0 - p1)), round to nearest even; the residuals are exact in fp32
__device__ __forceinline__ void split3(const float x, __bf16& p0, __bf16& p1, __bf16& p2) {
    p0 = (__bf16)x;
    const float r1 = x - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}

// Two fp16 pieces (round to nearest even; the residual is exact in fp32): 22 bits of the operand, fp16's range (|x| <= 65504; the
// second piece of a value below 0.125 is an fp16 denormal, which v_mfma_f32_16x16x32_f16 honours: tools/micro/f16_split.hip).
// `bad` collects values the format cannot hold (beyond the range, Inf, NaN).
__device__ __forceinline__ void split2h(const float x, _Float16& p0, _Float16& p1, bool& bad) {
    bad |= !(fabsf(x) <= 65504.f);
    p0 = (_Float16)x;
    p1 = (_Float16)(x - (float)p0);
}

// fp32 rows [m][c] -> fp16 [m][2][c]
// (d_rows / pitch: the matrix is `pitch`-row replicas handed over at their bound with *d_rows valid rows each -- what lies behind them
//  is uninitialised and must not raise the range flag)
__global__ void split2h_rows_kernel(const float* __restrict__ src, int64_t m, int c, _Float16* __restrict__ dst, int32_t* __restrict__ status,
                                    const int32_t* __restrict__ d_rows, int64_t pitch) {
    const int c8 = c >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * c8) return;
    const int64_t row = i / c8;
    if (d_rows != nullptr && row % pitch >= (int64_t)*d_rows) return;
    const int cb = (int)(i % c8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src + row * c + cb), b = *reinterpret_cast<const float4*>(src + row * c + cb + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 q0, q1;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 p0, p1;
        split2h(v[e], p0, p1, bad);
        q0[e] = p0; q1[e] = p1;
    }
    _Float16* d = dst + row * 2 * c + cb;
    *reinterpret_cast<f16x8*>(d) = q0;
    *reinterpret_cast<f16x8*>(d + c) = q1;
    if (bad && status) atomicOr(status, kStatusF16Range);
}

// packed weights of the two-piece kernel: the fragment order of pack_weights_bf16_kernel (spconv_bf16.hip) with planes = 2, fp16
// pieces of w * scale (scale: a power of two chosen by the caller so that the second pieces are normal numbers)
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ w, int k_vol, int c_in, int c_out, int nslab, float scale,
                                          _Float16* __restrict__ wp, int64_t total, int32_t* __restrict__ status) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t rest = idx >> 9;
    const int plane = (int)(rest % 2);
    rest /= 2;
    const int nt16 = c_out >> 4;
    const int nt = (int)(rest % nt16);
    rest /= nt16;
    const int slab = (int)(rest % nslab);
    const int k = (int)(rest / nslab);
    const int kin = 32 * slab + 8 * (lane >> 4) + i;
    const int col = 16 * nt + (lane & 15);
    const float v = kin < c_in ? scale * w[((int64_t)k * c_in + kin) * c_out + col] : 0.f;
    _Float16 p0, p1;
    bool bad = false;
    split2h(v, p0, p1, bad);
    wp[idx] = plane ? p1 : p0;
    if (bad && status) atomicOr(status, kStatusF16Range);
}

// fp32 rows [m][c] -> bf16 [m][3][c]
__global__ void split3_rows_kernel(const float* __restrict__ src, int64_t m, int c, __bf16* __restrict__ dst) {
    const int c8 = c >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m * c8) return;
    const int64_t row = i / c8;
    const int cb = (int)(i % c8) * 8;
    const float4 a = *reinterpret_cast<const float4*>(src + row * c + cb), b = *reinterpret_cast<const float4*>(src + row * c + cb + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bf16x8 q0, q1, q2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        __bf16 p0, p1, p2;
        split3(v[e], p0, p1, p2);
        q0[e] = p0; q1[e] = p1; q2[e] = p2;
    }
    __bf16* d = dst + row * 3 * c + cb;
    *reinterpret_cast<bf16x8*>(d) = q0;
    *reinterpret_cast<bf16x8*>(d + c) = q1;
    *reinterpret_cast<bf16x8*>(d + 2 * c) = q2;
}

// Sort keys for the rows of a neighbour table: one bit per offset = "the row has a neighbour under it"; the centre offset (every
// valid row has it) also as bit 27, so that a DESCENDING sort puts the valid rows of a table handed over at its bound first.  Rows
// sorted by this key sit next to rows with (nearly) the same set of offsets: whole 16-row blocks then lack an offset and the kernel
// below skips them (its block masks).  A sort makes about log2(rows / 16) ~ 12-13 leading key bits uniform within a block and leaves
// the others to chance, so the leading bits go to the offsets that are present LEAST often -- there a uniform block is most likely an
// empty one: on a scan's surfaces the offsets out of the horizontal plane (dz != 0), corners before edges before faces; the plane's
// own offsets and the centre last.  Against plain offset order: 11 % fewer executed blocks on the late steps' maps (sigma 0.05), 2-3 %
// at sigma 0.3, none on the early steps' near-random clouds (tools/mask_order_study.py, profiles/r06_mask_orders.txt).
// (offset k = (dx, dy, dz) with dx fastest: coords.hip kernel_map_self_kernel)
__constant__ int8_t kKeyBit27[27] = {26, 18, 25, 17, 10, 16, 24, 15, 23, 8, 4, 7, 3, 0, 2, 6, 1, 5, 22, 14, 21, 13, 9, 12, 20, 11, 19};
__global__ void row_mask_keys_kernel(const int32_t* __restrict__ nbr, int k_vol, int64_t m, int32_t* __restrict__ keys) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    int32_t key = 0;
    for (int k = 0; k < k_vol; ++k)
        if (nbr[(int64_t)k * m + r] >= 0) key |= 1 << (k_vol == 27 ? kKeyBit27[k] : k);
    if (nbr[(int64_t)(k_vol / 2) * m + r] >= 0) key |= 1 << 27;
    keys[r] = key;
}

#ifndef LIDIFF_S3_NRQ
#define LIDIFF_S3_NRQ 4
#endif
#ifndef LIDIFF_S3_CBW
#define LIDIFF_S3_CBW 2          // column blocks per wave on 128-column tiles (4 = 64 x 64 wave tiles: fewer LDS reads, measured +1.5 % SLOWER)
#endif
// NP = 3: bf16 pieces, six products (the default: fp32 accuracy).  NP = 2: fp16 pieces of 22-bit operands, three products -- half the
// matrix work, 4 instead of 6 bytes per gathered element; p.out_scale undoes the power of two the weights were packed with.
template <int BN, int NP = 3>
__global__ __launch_bounds__(512) void spconv_fwd_split3_kernel(const ConvParams p_launch) {
    using piece_t = typename std::conditional<NP == 3, __bf16, _Float16>::type;
    using frag_t = typename std::conditional<NP == 3, bf16x8, f16x8>::type;
    // 8 waves as RG row groups x CG column groups, a wave = RBW row blocks x 2 column blocks: 128 columns -> 2 x 4 (8 x 2 blocks),
    // 64 columns -> 4 x 2 (4 x 2 blocks)
    constexpr int BM = 256, KS = 32, NW = 8, CBW = (BN == 128 ? LIDIFF_S3_CBW : 2), CG = (BN / 16) / CBW, RG = NW / CG;
    constexpr int RBW = (BM / 16) / RG;                     // row blocks per wave (8 / 4)
    constexpr int APLANE = BM * KS * 2;                     // one piece of the stage's rows: 16 KB
    constexpr int ABYTES = NP * APLANE;
    constexpr int WBLK = NP * (BN / 16);                    // 1 KB W blocks per stage: [column block][piece]
    constexpr int WBYTES = WBLK * 1024;
    constexpr int STAGE = ABYTES + WBYTES;
    constexpr int NRQ = LIDIFF_S3_NRQ;                      // waves that issue requests (8: all; 4: the first wave of every SIMD only)
    constexpr int NCHK = 4, RPI = 16, T = (BM / RPI) / NRQ; // row requests per issuing wave, piece and stage
    static_assert((BM / RPI) % NRQ == 0 && RBW % 2 == 0 && CG * RG == NW && (BN == 128 || BN == 64), "request split");
    ConvParams p = p_launch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* act = reinterpret_cast<int32_t*>(smem + 2 * STAGE);          // offsets with a neighbour in the tile; act[31] = count
    int32_t* bmask = act + 32;                                            // per offset: which of the tile's 16-row blocks hold a neighbour
    int32_t* rowbuf = bmask + 32;                                         // [2][BM]: the tile's column of the table for an offset

    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tmr = (g / p.tiles_n) * 8 + xcd;
    const int64_t m_valid = valid_rows(p);
    const int tiles_live = p.d_m_out ? (int)((m_valid + BM - 1) / BM) : p.tiles_m;
    if (tmr >= tiles_live * p.replicas) return;
    const int rep = tmr / tiles_live, tm = tmr - rep * tiles_live;
    const int pitch_a = NP * p.c_in_a * 2, pitch_b = NP * p.c_in_b * 2;  // bytes per row of the split matrices
    const char* in_a = reinterpret_cast<const char*>(p.in_a) + (int64_t)rep * p.m_in * pitch_a;
    const char* in_b = p.in_b ? reinterpret_cast<const char*>(p.in_b) + (int64_t)rep * p.m_in * pitch_b : nullptr;
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    piece_t* out3 = p.out_planes ? reinterpret_cast<piece_t*>(p.out_planes) + (int64_t)rep * p.m_out * NP * p.c_out : nullptr;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, m_valid - row0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave / CG, cg = wave % CG;
    const int li = lane & 15, lq = lane >> 4;

    // ---- which offsets occur in the tile ---------------------------------------------------------------------------
    auto nbr_at = [&](int k, int r) -> int {
        if (r >= rows_here) return -1;
        return p.nbr ? p.nbr[(int64_t)k * p.m_out + row0 + r] : (int32_t)(row0 + r);
    };
    if (tid < 32) act[tid] = 0;
    __syncthreads();
    // ... and, per offset, which of the tile's sixteen 16-row blocks hold a neighbour at all: a block without one is neither
    // requested nor multiplied.  In table order that hardly ever happens on a dense level (P = (1 - occupancy)^16); with the
    // caller's rows sorted by their neighbour masks (row_order: lidiff_row_mask_keys + a sort) the rows that have offset k sit
    // together and 30-70 % of the (block, offset) pairs drop out.
    for (int k = wave; k < p.k_vol; k += NW) {
        int v[BM / 64];
#pragma unroll
        for (int c = 0; c < BM / 64; ++c) v[c] = nbr_at(k, 64 * c + lane);
        unsigned bm = 0;
#pragma unroll
        for (int c = 0; c < BM / 64; ++c) {
            const unsigned long long b = __ballot(v[c] >= 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((b >> (16 * q)) & 0xffffull) bm |= 1u << (4 * c + q);
        }
        if (lane == 0) { act[k] = bm != 0; bmask[k] = (int32_t)bm; }
    }
    __syncthreads();
    if (wave == 0) {
        const bool on = lane < p.k_vol && act[lane] != 0;
        const int mine = lane < p.k_vol ? bmask[lane] : 0;
        const unsigned long long m = __ballot(on);          // (one wave, in lockstep: every flag is read before any is overwritten)
        if (on) { act[popc_below(m)] = lane; }
        if (lane == 0) act[31] = __popcll(m);
        (void)mine;
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(act[31]);
    const int nslab = p.c_in / KS;
    const int nst = nact * nslab;
    const int nt16 = p.c_out >> 4;
    const int w_slab_bytes = nt16 * NP * 1024;
    auto rsrc = [](const void* base, int64_t bytes) {
        const uint64_t a = (uint64_t)(uintptr_t)base;
        i32x4 d = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_readfirstlane(d[i]);
        return d;
    };
    const i32x4 rsrc_w = rsrc(p.wp, (int64_t)p.k_vol * nslab * w_slab_bytes);
    const i32x4 rsrc_a = rsrc(in_a, p.m_in * (int64_t)pitch_a);
    const i32x4 rsrc_b = in_b ? rsrc(in_b, p.m_in * (int64_t)pitch_b) : rsrc_a;
    auto swz = [](int r) { return (r >> 2) & 2; };          // (ds_read_b128's lane groups: see spconv_fwd_bf16_kernel)
    int chb[T];
    const int rqw = wave % NRQ;                              // (waves behind NRQ mirror one of the issuing waves' bookkeeping and issue nothing)
#pragma unroll
    for (int j = 0; j < T; ++j) chb[j] = 16 * ((lane % NCHK) ^ swz(RPI * (rqw + NRQ * j) + lane / NCHK));
    const int foff = li * (KS * 2) + 16 * (lq ^ swz(li));
    const unsigned s_base = (unsigned)(uintptr_t)(lds_ptr_t)smem;

    // the gather rows of this lane's requests: those of the offset being requested, and (loaded one offset ahead) the next one's
    // The gather rows of this lane's requests.  An offset's column of the neighbour table (the tile's 256 entries, 1 KB) is itself
    // fetched by LDS-DMA one offset ahead -- four 256-byte requests, one per wave 0 .. 3 -- into rowbuf, and read from there when the
    // offset's first stage is requested: nothing the compiler tracks is in flight inside the loop (plain loads here made it put
    // `s_waitcnt vmcnt(0)` -- a wait for every request in flight -- in front of the requests of a stage).
    int row_cur[T];
    const i32x4 rsrc_n = p.nbr ? rsrc(p.nbr, (int64_t)p.k_vol * p.m_out * 4) : rsrc_a;
    const unsigned rowbuf_base = (unsigned)(uintptr_t)(lds_ptr_t)rowbuf;
    auto request_rows = [&](int oi) {                        // the table column of active offset oi -> rowbuf[oi & 1]
        if (p.nbr == nullptr || oi >= nact || wave >= BM / 64) return;
        const int k = to_sgpr(act[oi]);
        // (entries behind the tile's last row are masked when they are read; the descriptor's range check keeps the request
        // inside the table)
        const int64_t e0 = (int64_t)k * p.m_out + row0 + 64 * wave;
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                     :: "s"(rowbuf_base + (oi & 1) * (BM * 4) + wave * 256), "v"((int)(e0 * 4) + lane * 4), "s"(rsrc_n), "s"(0) : "memory");
    };
    auto take_rows = [&](int oi) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int r = RPI * (rqw + NRQ * j) + lane / NCHK;
            row_cur[j] = r >= rows_here ? -1 : p.nbr ? rowbuf[(oi & 1) * BM + r] : (int32_t)(row0 + r);
        }
    };
    // the requests of one stage, in two parts (placed separately in the wave's instruction stream, see the loop): next_stage()
    // fixes what they ask for -- (offset, slab), the gather rows -- and moves the cursor on
    int i_oi = 0, i_slab = 0;
    int n_slot = 0, n_ws = 0, n_cw = 0, n_cb = 0, n_k = 0;
    unsigned n_bm = 0;                                       // the requested stage's block mask
    bool n_from_a = true;
    auto next_stage = [&](int sg) {
        n_slot = sg & 1;
        if (i_slab == 0) {                                   // a new offset: its rows were requested one offset ago
            take_rows(i_oi);
            request_rows(i_oi + 1);
            n_k = to_sgpr(act[i_oi]);                        // (looked up once per offset, not per stage: two LDS round trips at the
            n_bm = (unsigned)to_sgpr(bmask[n_k]);            //  head of a stage, in front of its requests)
        }
        const int k = n_k;
        n_ws = (k * nslab + i_slab) * w_slab_bytes;
        const int k0 = i_slab * KS;
        n_from_a = k0 < p.c_in_a;
        n_cw = n_from_a ? p.c_in_a * 2 : p.c_in_b * 2;       // bytes of one piece of a row
        n_cb = (n_from_a ? k0 : k0 - p.c_in_a) * 2;
        if (++i_slab == nslab) { i_slab = 0; ++i_oi; }
    };
    auto issue_w = [&]() {
#pragma unroll
        for (int b = rqw; b < WBLK; b += NRQ)
            s3_dma16(rsrc_w, s_base + n_slot * STAGE + ABYTES + b * 1024, (((n0 >> 4) * NP + b) * 64 + lane) * 16, n_ws);
    };
    auto issue_a = [&]() {
        int soff[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) soff[q] = n_cb + q * n_cw;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int t = rqw + NRQ * j;                     // a request = one 16-row block of one piece
            if (!((n_bm >> t) & 1u)) continue;               // (no neighbour in the block under this offset: not requested, not multiplied)
            const int voff = row_cur[j] >= 0 ? row_cur[j] * (NP * n_cw) + chb[j] : (int)0x80000000;   // no neighbour: zeros, no bytes moved
#pragma unroll
            for (int q = 0; q < NP; ++q)
                s3_dma16(n_from_a ? rsrc_a : rsrc_b, s_base + n_slot * STAGE + q * APLANE + t * 1024, voff, soff[q]);
        }
    };
#define LIDIFF_S3_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    f32x4 acc[RBW][CBW];
#pragma unroll
    for (int j = 0; j < RBW; ++j)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Two levels of fp32 sums, as in the native kernel (whose pair lists sum every offset on its own before the tile adds them up):
    // acc_k over the channels and pieces of ONE offset (8 C_in / 32 x 6 MFMAs), acc over the offsets.  One chain over everything --
    // 27 x C_in / 32 x 6 accumulations -- measured 8x the native kernel's error against float64 (4.0e-5 vs 5.2e-6 on outputs of
    // scale 22: the rounding of a long fp32 chain, not the split); the sums here are as long as the native kernel's.
    f32x4 acc_k[RBW][CBW];
#pragma unroll
    for (int j = 0; j < RBW; ++j)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc_k[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    request_rows(0);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (nst > 0) {
        next_stage(0);
        if (wave < NRQ) { issue_a(); issue_w(); }
    }
    int c_slab = 0;
#ifdef LIDIFF_CONV_PROBE
    const int abl = p.flags;                                 // probe build only (LIDIFF_S3_ABLATE; results are wrong with any bit set):
#else                                                        // 1 no requests, 2 no MFMAs, 4 no barrier, 8 / 16 no A / W fragment reads
    constexpr int abl = 0;
#endif
#ifdef LIDIFF_CONV_PROBE
    long long tq_bar = 0, tq_head = 0, tq_dma = 0, tq_w = 0, tq_mma = 0, tq_fold = 0;
    const long long tq_start = __builtin_readcyclecounter();
#define S3_T(var) { const long long now_ = __builtin_readcyclecounter(); var += now_ - tq_last; tq_last = now_; }
#else
#define S3_T(var)
#endif
    for (int sg = 0; sg < nst; ++sg) {
#ifdef LIDIFF_CONV_PROBE
        long long tq_last = __builtin_readcyclecounter();
#endif
        if (!(abl & 4)) LIDIFF_S3_BARRIER();                 // stage sg has landed for every wave; the other slot is free
        S3_T(tq_bar);
        const bool more = sg + 1 < nst;
        const unsigned c_bm = n_bm;                          // the block mask of the stage being multiplied
        // The requests of the next stage: the first row group issues them in front of its MFMAs, the second one -- the SIMD
        // partners -- in the middle of its own: while one wave of a SIMD spends its ~10 request slots the other one multiplies
        // (measured, 256 -> 256 at stride 8: 3.55 ms against 3.69 with all eight waves requesting at the head of the stage; where
        // exactly the second group places them -- after one, two or three quarters of its MFMAs, rows and W apart or together --
        // moves nothing: +-2 %).  What bounds the kernel is the matrix pipe under the chip's power limit: 1.39 PFLOP/s executed,
        // the range MI355X_MICROARCH.md quotes for tuned bf16 attention (1.25-1.48) -- with all gather traffic switched off the
        // same launch takes 3.33 ms.  So the lever is the NUMBER of MFMAs: the block masks.
        if (more) next_stage(sg + 1);
        S3_T(tq_head);
        if (more && wave < NW / 2 && wave < NRQ && !(abl & 1)) { issue_a(); issue_w(); }
        S3_T(tq_dma);
        const char* st = smem + (sg & 1) * STAGE;
        const char* wsrc = st + ABYTES + (CBW * cg) * NP * 1024 + lane * 16;
        // this wave's row blocks: RG j + rg, j = 0 .. RBW - 1 (interleaved between the row groups: under sorted rows the blocks that
        // hold an offset are neighbours, and every wave should get its share of them)
        const char* asrc = st + rg * (16 * KS * 2) + foff;
        frag_t w[CBW][NP];
        if (!(abl & 16)) {
#pragma unroll
            for (int c = 0; c < CBW; ++c)
#pragma unroll
                for (int q = 0; q < NP; ++q) w[c][q] = *reinterpret_cast<const frag_t*>(wsrc + (c * NP + q) * 1024);
        }
        // Row block after row block: the fragment reads of block j + 1 are issued (always -- a block that is skipped costs three
        // idle LDS reads) in front of the twelve MFMAs of block j (only if the block holds a neighbour under this offset: a
        // wave-uniform branch), so that no MFMA group waits for its own reads.  The six products, smallest first; swapped operands
        // (W fragment first): a lane ends up with 4 channels of one row.
#define LIDIFF_S3_PRODUCT(J, A, QA, QW)                                                                                    \
    _Pragma("unroll") for (int c = 0; c < CBW; ++c)                                                                        \
        acc_k[J][c] = s3_mfma(w[c][QW], A[QA], acc_k[J][c])
        auto read_block = [&](int j, frag_t* a) {
            if (abl & 8) return;
#pragma unroll
            for (int q = 0; q < NP; ++q) a[q] = *reinterpret_cast<const frag_t*>(asrc + q * APLANE + (RG * j) * (16 * KS * 2));
        };
        frag_t fa[2][NP];
        auto block = [&](auto j_tag) {
            constexpr int J = decltype(j_tag)::value;
            if constexpr (J + 1 < RBW) read_block(J + 1, fa[(J + 1) & 1]);
            if (((c_bm >> (RG * J + rg)) & 1u) && !(abl & 2)) {
                if constexpr (NP == 3) {
                    LIDIFF_S3_PRODUCT(J, fa[J & 1], 2, 0); LIDIFF_S3_PRODUCT(J, fa[J & 1], 0, 2); LIDIFF_S3_PRODUCT(J, fa[J & 1], 1, 1);
                }
                LIDIFF_S3_PRODUCT(J, fa[J & 1], 1, 0); LIDIFF_S3_PRODUCT(J, fa[J & 1], 0, 1); LIDIFF_S3_PRODUCT(J, fa[J & 1], 0, 0);
            }
        };
        S3_T(tq_w);
        read_block(0, fa[0]);
        block(ic<0>{});
        block(ic<1>{});
        if constexpr (RBW >= 8) { block(ic<2>{}); block(ic<3>{}); }
        S3_T(tq_mma);
        if (more && wave >= NW / 2 && wave < NRQ && !(abl & 1)) { issue_a(); issue_w(); }
        S3_T(tq_dma);
        if constexpr (RBW >= 8) { block(ic<4>{}); block(ic<5>{}); block(ic<6>{}); block(ic<7>{}); }
        else { block(ic<2>{}); block(ic<3>{}); }
        S3_T(tq_mma);
#undef LIDIFF_S3_PRODUCT
        if (++c_slab == nslab) {                             // the offset is complete: its sums join the others'
            c_slab = 0;
#pragma unroll
            for (int j = 0; j < RBW; ++j)
#pragma unroll
                for (int c = 0; c < CBW; ++c) {
                    acc[j][c] += acc_k[j][c];
                    acc_k[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        S3_T(tq_fold);
    }
#undef LIDIFF_S3_BARRIER
#ifdef LIDIFF_CONV_PROBE
    if (p.timeline != nullptr && lane == 0 && (wave == 0 || wave == NW / 2)) {
        long long* d = p.timeline + ((int64_t)bid * 2 + (wave != 0)) * 10;
        d[0] = __builtin_readcyclecounter() - tq_start; d[1] = tq_bar; d[2] = tq_head; d[3] = tq_dma; d[4] = tq_w; d[5] = tq_mma;
        d[6] = tq_fold; d[7] = nst; d[8] = tq_start - __builtin_readcyclecounter(); d[9] = 1;
    }
#endif

    // ---- epilogue straight from the registers: lane (li, lq) holds channels n0 + 16 (CBW cg + c) + 4 lq .. + 3 of tile row 16 (RG j + rg) + li
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
        const int col = n0 + 16 * (CBW * cg + c) + 4 * lq;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + col);
#pragma unroll
        for (int j = 0; j < RBW; ++j) {
            const int r = 16 * (RG * j + rg) + li;
            if (r >= rows_here) continue;
            const int64_t orow = p.row_order ? (int64_t)p.row_order[row0 + r] : row0 + r;     // tile row -> output row
            const int64_t o = orow * p.c_out + col;
            float4 v = make_float4(acc[j][c][0], acc[j][c][1], acc[j][c][2], acc[j][c][3]);
            if constexpr (NP == 2) { v.x *= p.out_scale; v.y *= p.out_scale; v.z *= p.out_scale; v.w *= p.out_scale; }   // (a power of two: exact)
            if (p.scale) { v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            if (p.shift) { v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w; }
            if (p.residual) {
                const float4 s = *reinterpret_cast<const float4*>(p.residual + o);
                v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
            }
            if (p.relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<float4*>(p.out + o) = v;
            if (out3) {                                      // the next dense convolution's operand, cut here
                const float vv[4] = {v.x, v.y, v.z, v.w};
                if constexpr (NP == 3) {
                    bf16x4 q0, q1, q2;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        __bf16 p0, p1, p2;
                        split3(vv[e], p0, p1, p2);
                        q0[e] = p0; q1[e] = p1; q2[e] = p2;
                    }
                    __bf16* d = out3 + orow * 3 * p.c_out + col;
                    *reinterpret_cast<bf16x4*>(d) = q0;
                    *reinterpret_cast<bf16x4*>(d + p.c_out) = q1;
                    *reinterpret_cast<bf16x4*>(d + 2 * p.c_out) = q2;
                } else {
                    f16x4 q0, q1;
                    bool bad = false;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        _Float16 p0, p1;
                        split2h(vv[e], p0, p1, bad);
                        q0[e] = p0; q1[e] = p1;
                    }
                    _Float16* d = out3 + orow * 2 * p.c_out + col;
                    *reinterpret_cast<f16x4*>(d) = q0;
                    *reinterpret_cast<f16x4*>(d + p.c_out) = q1;
                    if (bad && p.status) atomicOr(p.status, kStatusF16Range);
                }
            }
        }
    }
}

template <int BN, int NP>
static int launch_split3(const ConvParams& p, hipStream_t st) {
    constexpr size_t lds = 2 * (size_t)(NP * 256 * 32 * 2 + NP * (BN / 16) * 1024) + 2 * 32 * 4 + 2 * 256 * 4;
    auto kern = spconv_fwd_split3_kernel<BN, NP>;
    static thread_local bool configured = false;
    if (!configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, 256);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

}  // namespace lidiff

using namespace lidiff;

#ifdef LIDIFF_CONV_PROBE
static long long* g_s3_timeline = nullptr;
extern "C" void lidiff_debug_set_split3_timeline(long long* buf) { g_s3_timeline = buf; }
#endif

extern "C" int lidiff_split3_rows(const float* src, int64_t m, int32_t c, int32_t pieces, void* dst, int32_t* d_status,
                                  const int32_t* d_rows, int64_t pitch, void* stream) {
    LIDIFF_CHECK_ARG(m >= 0 && c > 0 && c % 8 == 0, "rows >= 0, channels a multiple of 8");
    LIDIFF_CHECK_ARG(pieces == 3 || pieces == 2, "pieces: 3 (bf16) or 2 (fp16)");
    if (m == 0) return 0;
    LIDIFF_CHECK_ARG(src != nullptr && dst != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pointers must be 16-byte aligned");
    const unsigned grid = (unsigned)ceil_div(m * (c / 8), 256);
    if (pieces == 3) split3_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, m, c, (__bf16*)dst);
    else {
        LIDIFF_CHECK_ARG(d_rows == nullptr || (pitch > 0 && m % pitch == 0), "d_rows: the matrix must be whole replicas of `pitch` rows");
        split2h_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, m, c, (_Float16*)dst, d_status, d_rows, pitch);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_spconv_pack_weights_f16x2(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out, float scale, void* w_packed,
                                                int32_t* d_status, void* stream) {
    LIDIFF_CHECK_ARG(w != nullptr && w_packed != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && c_in > 0, "kernel volume must be 1..27, c_in > 0");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    LIDIFF_CHECK_ARG(scale > 0.f && frexpf(scale, (int[1]){0}) == 0.5f, "scale must be a power of two");
    const int nslab = (c_in + 31) / 32;
    const int64_t total = (int64_t)k_vol * nslab * 32 * c_out * 2;
    pack_weights_f16x2_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w, k_vol, c_in, c_out, nslab, scale,
                                                                                              (_Float16*)w_packed, total, d_status);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_row_mask_keys(const int32_t* nbr, int32_t k_vol, int64_t m, int32_t* keys, void* stream) {
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && m >= 0, "kernel volume 1..27, rows >= 0");
    if (m == 0) return 0;
    LIDIFF_CHECK_ARG(nbr != nullptr && keys != nullptr, "null pointer");
    row_mask_keys_kernel<<<(unsigned)ceil_div(m, 256), 256, 0, (hipStream_t)stream>>>(nbr, k_vol, m, keys);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t lidiff_spconv_fwd_split3_supported(int32_t c_in_a, int32_t c_in_b, int32_t c_out) {
    return c_in_a > 0 && c_in_a % 32 == 0 && c_in_b >= 0 && c_in_b % 32 == 0 && c_out > 0 && c_out % 64 == 0;
}

extern "C" int lidiff_spconv_fwd_split3(const void* in_a3, int32_t c_in_a, const void* in_b3, int32_t c_in_b, const void* w_packed3,
                                        const int32_t* nbr, int32_t k_vol, int64_t m_in, int64_t m_out, int32_t c_out, float* out,
                                        void* out_planes, const float* ep_scale, const float* ep_shift, const float* residual,
                                        int32_t relu, int32_t replicas, const int32_t* d_m_out, const int32_t* row_order, int32_t pieces,
                                        float out_scale, int32_t* d_status, void* stream) {
    LIDIFF_CHECK_ARG(in_a3 != nullptr && w_packed3 != nullptr && out != nullptr, "null pointer");
    LIDIFF_CHECK_ARG((in_b3 == nullptr) == (c_in_b == 0), "in_b3 and c_in_b must agree");
    LIDIFF_CHECK_ARG(lidiff_spconv_fwd_split3_supported(c_in_a, c_in_b, c_out), "widths: inputs multiples of 32, c_out a multiple of 64");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(nbr != nullptr || (k_vol == 1 && m_in == m_out), "identity map needs K=1, m_in==m_out");
    LIDIFF_CHECK_ARG(replicas >= 1 && m_out >= 0 && m_in >= 0, "bad shape");
    LIDIFF_CHECK_ARG(pieces == 3 || (pieces == 2 && out_scale > 0.f), "pieces: 3 (bf16), or 2 (fp16) with the inverse of the weights' scale");
    if (m_out == 0) return 0;
    LIDIFF_CHECK_ARG(m_in > 0, "outputs without inputs");
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(in_a3) && al16(in_b3) && al16(w_packed3) && al16(out) && al16(out_planes) && al16(ep_scale) && al16(ep_shift) &&
                         al16(residual), "pointers must be 16-byte aligned");
    LIDIFF_CHECK_ARG(m_in * (int64_t)c_in_a * 2 * pieces < (1ll << 31) && m_in * (int64_t)c_in_b * 2 * pieces < (1ll << 31),
                     "a split feature matrix exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(nbr == nullptr || (int64_t)k_vol * m_out * 4 < (1ll << 31), "the neighbour table exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(lidiff_spconv_packed_weight_bf16_elems(k_vol, c_in_a + c_in_b, c_out, pieces) * 2 < (1ll << 31),
                     "packed weights exceed the 2 GiB buffer-descriptor range");
    ConvParams p{};
    p.in_a = reinterpret_cast<const float*>(in_a3); p.in_b = reinterpret_cast<const float*>(in_b3);
    p.wp = reinterpret_cast<const float*>(w_packed3); p.nbr = nbr; p.row_order = row_order; p.out = out; p.out_planes = out_planes;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.m_in = m_in; p.m_out = m_out; p.d_m_out = d_m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu; p.replicas = replicas; p.out_scale = out_scale; p.status = d_status;
#ifdef LIDIFF_CONV_PROBE
    p.timeline = g_s3_timeline;
    { static const int abl = [] { const char* e = getenv("LIDIFF_S3_ABLATE"); return e ? atoi(e) : 0; }(); p.flags = abl; }
#endif
    if (pieces == 2) return c_out % 128 == 0 ? launch_split3<128, 2>(p, (hipStream_t)stream) : launch_split3<64, 2>(p, (hipStream_t)stream);
    return c_out % 128 == 0 ? launch_split3<128, 3>(p, (hipStream_t)stream) : launch_split3<64, 3>(p, (hipStream_t)stream);
}
