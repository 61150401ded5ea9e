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
//   * B operand: the wave's [KS k x 16 col] piece of W[k] comes straight from HBM/L2 into VGPRs in
//     MFMA fragment order (weights are pre-packed once by lidiff_spconv_pack_weights: fully
//     coalesced 2 KB runs) and is reused by all of the wave's row blocks.
//   * A operand: the gathered rows of a stage (<= 128 pairs x KS channels) are written by LDS-DMA
//     (buffer_load_dwordx4 ... lds: no staging VGPRs, no ds_write pass) into a double-buffered
//     [128][KS] image whose 16-byte chunks are XOR-swizzled THROUGH THE SOURCE ADDRESS, which makes
//     the ds_read_b128 fragment reads of all waves bank-conflict free.  K is consumed in a permuted
//     order shared by both operands.
//   * A stage = (offset, chunk of <= 128 pairs, KS-channel slab), KS = 64 when the channel counts
//     allow it (half the barriers and half the per-stage control per MFMA), else 32.  Loads of stage
//     i+1 are issued before the MFMAs of stage i; one barrier per stage.
//   * After the last slab of an offset the register accumulators are added into the LDS tile through
//     the pair list's local output row (each output row occurs at most once per offset and waves own
//     disjoint column / row-block sets, so the LDS read-modify-write is race free).
//
// Replaces ME's ConvolutionForwardGPU (gather -> GEMM -> atomic scatter per offset) behind
// MinkowskiConvolution / MinkowskiConvolutionTranspose; call sites in include/lidiff_amd.h.
#include <type_traits>

#include "common.h"
#include "spconv.h"

namespace lidiff {

template <int BM, int WN, int WM, int KS>
struct ConvCfg {
    static constexpr int BN = 16 * WN;
    static constexpr int NW = WN * WM;
    static constexpr int NT = 64 * NW;
    static constexpr int RB = kChunk / 16;       // row blocks per stage
    static constexpr int RBW = RB / WM;          // row blocks per wave
    static constexpr int NCH = BM / kChunk;      // chunks per offset (upper bound)
    static constexpr int A_FLOATS = kChunk * KS; // one A image (16 / 32 KB)
    static constexpr int WORK_INTS = 27 * (BM / 16) + 8;   // work list capacity: 27 offsets x up to BM / 16 segments of 16 rows
    static_assert(KS == 32 || KS == 64, "KS");
    static_assert(BM % kChunk == 0, "BM");
    static_assert(RB % WM == 0, "WM must divide 8");
    static_assert(NT <= 1024, "workgroup size");

    // pair list of output rows: byte offsets into the accumulator tile (dummy row included); 16-bit where they
    // fit -- the narrow tiles of the low-density maps stay below 80 KB of LDS, i.e. two workgroups per CU
    using OutT = std::conditional_t<((BM + 1) * BN * 4 < 65536), uint16_t, int32_t>;
    __host__ __device__ static size_t lds_bytes(int k_vol) {
        size_t b = 2 * (size_t)A_FLOATS * 4;              // A images (LDS-DMA targets, kept below 64 KB)
        b += (size_t)(BM + 1) * BN * 4;                   // accumulator tile + one dummy row (branch-free flush)
        b += (size_t)k_vol * BM * 4;                      // in_list
        b += 32 * 4;                                      // cnt (k_vol <= 27; cnt[31] = #work items)
        b += (size_t)WORK_INTS * 4;                       // work list
        b += (size_t)BM * 4;                              // output row of every tile row
        b += (size_t)k_vol * BM * sizeof(OutT);           // out_list
        return (b + 15) & ~(size_t)15;
    }
};

// One tile slot `bid` of the launch: everything from the pair lists to the epilogue.
template <int BM, int WN, int WM, int KS, bool VEC>
__device__ __forceinline__ void conv_tile(const ConvParams& p_launch, const int bid) {
    using Cfg = ConvCfg<BM, WN, WM, KS>;
    ConvParams p = p_launch;                     // per-replica view (pointers moved below)
    constexpr int BN = Cfg::BN, NT = Cfg::NT, NW = Cfg::NW, RBW = Cfg::RBW, AF = Cfg::A_FLOATS;
    constexpr int NJ = KS / 16;                  // 16-channel MFMA groups (4 MFMAs each) per stage
    constexpr int NCHK = KS / 4;                 // 16-byte chunks per image row
    constexpr int RPI = 64 / NCHK;               // image rows per LDS-DMA wave-instruction (1 KB)
    constexpr int NINST = kChunk / RPI;          // DMA instructions per full stage
    constexpr int T = (NINST + NW - 1) / NW;     // ... per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a_buf = reinterpret_cast<float*>(smem);
    float* acc_lds = a_buf + 2 * AF;
    int32_t* in_list = reinterpret_cast<int32_t*>(acc_lds + (BM + 1) * BN);
    int32_t* cnt = in_list + p.k_vol * BM;
    int32_t* work = cnt + 32;
    int32_t* orow = work + Cfg::WORK_INTS;                    // output row of every tile row
    // out_list[k][q] = byte offset of the accumulator row of pair q of offset k (its tile row x BN x 4).
    // Entries behind an offset's last pair point at the dummy row BM, so the flush needs no bounds test
    // and one add per element as its only address arithmetic.
    using OutT = typename Cfg::OutT;
    OutT* out_list = reinterpret_cast<OutT*>(orow + BM);
    auto out4 = [&](const OutT* q) {                          // four consecutive list entries (16- or 8-byte read)
        if constexpr (sizeof(OutT) == 4) {
            return *reinterpret_cast<const int4*>(q);
        } else {
            const uint2 w = *reinterpret_cast<const uint2*>(q);
            return make_int4((int)(w.x & 0xffff), (int)(w.x >> 16), (int)(w.y & 0xffff), (int)(w.y >> 16));
        }
    };
    constexpr int kDummyRow = BM * BN * 4;
    // The accumulator tile is addressed in 16-byte chunks (4 consecutive channels of one row -- what a lane holds of a block's
    // MFMA result, see run_item) and a row's chunks are XOR-permuted by the row's low bits: the 16 lanes of a quarter wave flush
    // 16 DIFFERENT rows at the same channel offset, and with the plain layout (row pitch = a multiple of all 64 banks) they would
    // all hit the same four banks.  SWZ = the largest power of two dividing the chunks per row, at most 16.
    constexpr int NCHR = BN / 4;
    constexpr int SWZ = (NCHR % 16 == 0) ? 16 : (NCHR % 8 == 0) ? 8 : (NCHR % 4 == 0) ? 4 : (NCHR % 2 == 0) ? 2 : 1;
    auto tile_addr = [&](int row_byte_off, int chunk) {       // byte address of `chunk` of the tile row at row_byte_off
        return row_byte_off + 16 * (chunk ^ ((row_byte_off / (BN * 4)) & (SWZ - 1)));
    };

    // XCD-aware tile mapping: the column tiles of one row tile share an XCD (their gathers hit the
    // same L2), consecutive row tiles round-robin over the 8 XCDs.
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tmr = (g / p.tiles_n) * 8 + xcd;  // row tile over all replicas
    const int64_t m_valid = valid_rows(p);       // == m_out unless the row count lives on the device (m_out: bound and pitch)
    // Replicas: the same kernel map and weights applied to `replicas` stacked feature matrices (the
    // conditional / unconditional pair of classifier-free guidance, pipeline:148-153): one launch, twice the tiles,
    // replica after replica (interleaving the replicas' tiles costs 1-2.7 % on the dense layers: measured).  With a
    // device-side row count below its bound only the LIVE row tiles are numbered, so every slot behind them -- those
    // workgroups leave at once -- sits at the end of the grid, not between the replicas (where a launch at 64 % of its
    // bound lost 1 %).
    const int tiles_live = p.d_m_out ? (int)((m_valid + BM - 1) / BM) : p.tiles_m;
    if (tmr >= tiles_live * p.replicas) return;
    const int rep = tmr / tiles_live, tm = tmr - rep * tiles_live;
    p.in_a += (int64_t)rep * p.m_in * p.c_in_a;
    if (p.in_b) p.in_b += (int64_t)rep * p.m_in * p.c_in_b;
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    if (p.tail) p.tail += (int64_t)rep * p.tail_rows * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, m_valid - row0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    const int wn = wave % WN, wm = wave / WN;
    STAMP(t_start);
#ifdef LIDIFF_CONV_PROBE
    const long long rt_start = __builtin_amdgcn_s_memrealtime();      // 100 MHz wall clock
    long long t_barrier = 0, t_flush = 0, t_issue = 0, t_mma = 0;
#endif

    // ---- pair lists: ordered compaction of nbr[k, row0 : row0+rows_here] per offset --------
    auto zero_tile = [&]() {
        for (int e = tid; e < (BM + 1) * BN / 4; e += NT)     // the tile and the dummy row behind it
            reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = tid; r < rows_here; r += NT) orow[r] = p.row_order ? p.row_order[row0 + r] : (int32_t)(row0 + r);
    };
    if (p.nbr == nullptr) {                      // kernel_size == 1: identity map
        zero_tile();
        for (int r = tid; r < BM; r += NT) {
            const int64_t gr = min(row0 + r, m_valid - 1);
            in_list[r] = p.row_order ? p.row_order[gr] : (int32_t)gr;
            out_list[r] = (OutT)(r < rows_here ? r * BN * 4 : kDummyRow);
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        // the tile's [k_vol x BM] block of the table goes through LDS first (A images are idle here):
        // every load of the block is in flight at once -- ONE global latency instead of one per offset.
        int32_t* raw = reinterpret_cast<int32_t*>(a_buf);
        static_assert(27 * BM * 4 <= 2 * AF * 4, "raw neighbour block must fit in the A images");
        // (all of a thread's loads are issued before the first is stored: the loop form left them to the compiler, which waited
        // for each -- a chain of up to 14 global latencies at the head of every tile)
        constexpr int NRAW = (27 * BM + NT - 1) / NT;
        int rawv[NRAW];
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int e = tid + i * NT, k = e / BM, r = e % BM;
            rawv[i] = (k < p.k_vol && r < rows_here) ? p.nbr[(int64_t)k * p.m_out + row0 + r] : -1;
        }
        zero_tile();                             // ... and the tile is zeroed while they are in flight
#pragma unroll
        for (int i = 0; i < NRAW; ++i) {
            const int e = tid + i * NT;
            if (e < p.k_vol * BM) raw[e] = rawv[i];
        }
        __syncthreads();
        for (int k = wave; k < p.k_vol; k += NW) {
            int pos = 0;
#pragma unroll
            for (int c = 0; c < BM; c += 64) {
                const int r = c + lane;
                const int v = raw[k * BM + r];
                const bool valid = v >= 0;
                const unsigned long long m = __ballot(valid);
                if (valid) {
                    const int q = pos + popc_below(m);
                    in_list[k * BM + q] = v;
                    out_list[k * BM + q] = (OutT)(r * BN * 4);
                }
                pos += __popcll(m);
            }
#pragma unroll
            for (int c = 0; c < BM; c += 64)
                if (c + lane >= pos) out_list[k * BM + c + lane] = (OutT)kDummyRow;
            if (lane == 0) cnt[k] = pos;
        }
    }
    __syncthreads();
    // ---- low-density tiles pack several offsets into one stage ---------------------------------
    // A stage multiplies up to 128 pair rows.  Where an offset brings only a handful of pairs per tile
    // (stride-1/2 levels: ~1-2 neighbours per voxel) a stage per offset is almost empty and the tile is
    // bound by per-stage latency, so such tiles split the stage into SEG segments of 128/SEG rows, each
    // with its own offset (own W fragment registers).  Decided per tile from its own pair counts.
    constexpr int SEG = (VEC && KS == 32) ? (WM == 1 ? 4 : 8) : 1;   // segments per stage in packed mode
    constexpr int SEGR = kChunk / SEG, BPS = SEGR / 16;               // rows / row blocks per segment
    bool packed = false;
    if constexpr (SEG > 1) {
        // stages per channel slab: one per active offset unpacked, ceil(#segments / SEG) packed (an offset with n pairs
        // takes ceil(n / SEGR) segments).  A packed stage costs about 4/3 of a plain one (W registers per segment,
        // ordered flush), so pack when that still wins -- e.g. the stride-1 level of a noisy scan: the centre offset
        // (128 pairs = SEG segments) plus ~7 offsets with one pair each are 2 packed stages instead of 8 plain ones
        // (the former rule, mean pairs per active offset <= 3/4 segment, kept such tiles unpacked: 21 of their 25
        // stages multiplied a single pair, profiles/r02_lowdensity_timeline.txt).
        const int c = lane < p.k_vol ? cnt[lane] : 0;
        int nseg = (c + SEGR - 1) / SEGR, act = c > 0;
        for (int off = 32; off > 0; off >>= 1) { nseg += __shfl_down(nseg, off); act += __shfl_down(act, off); }
        nseg = __builtin_amdgcn_readfirstlane(nseg);
        act = __builtin_amdgcn_readfirstlane(act);
        packed = p.nbr != nullptr && 4 * ((nseg + SEG - 1) / SEG) <= 3 * act;
    }
    const int segr = packed ? SEGR : kChunk;
    // ---- work list: (offset, segment of <= segr pairs), ascending k: wave 0, exclusive scan ------
    if (wave == 0) {
        const int c = lane < p.k_vol ? cnt[lane] : 0;
        const int nc = (c + segr - 1) / segr;
        int incl = nc;
        for (int off = 1; off < 32; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        for (int j = 0; j < nc; ++j)
            work[incl - nc + j] = lane | (j << 8) | (min(segr, c - j * segr) << 16);
        if (lane == 31) cnt[31] = incl;
    }
    __syncthreads();
    const int nwork = __builtin_amdgcn_readfirstlane(cnt[31]);
    const int nslab = (p.c_in + KS - 1) / KS;            // stages per work item
    const int nslab32 = (p.c_in + 31) / 32;              // 32-channel slabs of the packed weights
    const int nt16 = p.c_out >> 4;
    const int li = lane & 15, lq = lane >> 4;

    // ---- per-item state ---------------------------------------------------------------------
    struct Item { int k, start, n; };
    auto load_item = [&](int wi) {
        const int w = __builtin_amdgcn_readfirstlane(work[min(wi, nwork - 1)]);
        return Item{w & 0xff, ((w >> 8) & 0xff) * kChunk, w >> 16};
    };
    auto swz = [](int r) { return KS == 32 ? (r >> 1) & 7 : r & 15; };
    // fp32 MFMAs monopolise the SIMD's VALU (tools/micro/mfma_valu_share.hip: a partner wave's VALU work does
    // not overlap with them at all), so the slab loop keeps per-stage VALU work near zero: every per-lane
    // address is computed once per work item and the per-stage part of it travels in SGPRs (soffset).
    int rowv[T];                                          // source row of this lane's 16-byte piece (-1: none)
    int rowoff[T];                                        // its byte offset inside the current source (a or b), or OOB
    int chb[T];                                           // byte offset of the source chunk inside a slab row
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int r = RPI * (wave + NW * j) + lane / NCHK;
        chb[j] = 16 * ((lane % NCHK) ^ swz(r));
    }
    int list_base = 0;                                    // in_list offset of the item being gathered
    int rows_cw4 = 0;                                     // row pitch (bytes) rowoff[] was computed for
    auto set_rowoff = [&](int cw4) {
        rows_cw4 = cw4;
#pragma unroll
        for (int j = 0; j < T; ++j) rowoff[j] = rowv[j] >= 0 ? rowv[j] * cw4 + chb[j] : (int)0x80000000;   // OOB -> zeros
    };
    auto load_rows = [&](const Item& it) {
        list_base = it.k * BM + it.start;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int r = RPI * (wave + NW * j) + lane / NCHK;
                rowv[j] = (r < it.n && (T * NW == NINST || wave + NW * j < NINST)) ? in_list[list_base + r] : -1;
            }
            set_rowoff(p.c_in_a * 4);
        }
    };
    const int w_slab_bytes = nt16 * 512 * 4;              // bytes between consecutive 32-slabs of one offset
    const int w_lane_off = (((n0 >> 4) + wn) * 512 + lane * 4) * 4;   // this wave's fragment piece, per lane

    // Loads of one stage: A by LDS-DMA into the image at byte offset `img`, this wave's W fragment of
    // (offset k, slab) into w[].  `n` = pair rows of the stage's item, whose rows are in rowv[].
    auto issue = [&](int img, int k, int slab, int n, f32x4* w) {
        if (!PROBE(2)) {
            __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.wp), 0, (int)((size_t)p.k_vol * nslab32 * 32 * p.c_out * 4), 0x00020000);
            const int ws = (k * nslab32 + slab * (KS / 32)) * w_slab_bytes;       // wave-uniform: SGPR offset
#pragma unroll
            for (int h = 0; h < KS / 32; ++h) {
                w[2 * h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_lane_off, ws + h * w_slab_bytes, 0));
                w[2 * h + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_lane_off + 1024, ws + h * w_slab_bytes, 0));
            }
        }
        const int k0 = slab * KS;
        if constexpr (VEC) {
            const bool from_a = k0 < p.c_in_a;                 // uniform: slabs never straddle a|b
            const float* src = from_a ? p.in_a : p.in_b;
            const int cw4 = (from_a ? p.c_in_a : p.c_in_b) * 4;
            const int cb4 = (from_a ? k0 : k0 - p.c_in_a) * 4;
            if (cw4 != rows_cw4) set_rowoff(cw4);              // source changed (a <-> b): new row pitch
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(src), 0, (int)(p.m_in * cw4), 0x00020000);
            char* dst = reinterpret_cast<char*>(a_buf) + img;
#pragma unroll
            for (int j = 0; j < T; ++j) {                      // RPI rows x KS*4 B per wave-instruction; no branches,
                const int t = wave + NW * j;                   // no per-stage VALU: the slab offset rides in soffset
                // (an instruction whose RPI rows all lie behind the item's last pair is not issued: those image rows feed only
                // accumulator rows that are flushed into the dummy row, and every vector-memory instruction costs the SIMD
                // ~100 cycles of issue while fp32 MFMAs are queued -- wave-uniform test, scalar branch)
                if ((T * NW == NINST || t < NINST) && RPI * t < n && !PROBE(1)) {
                    const int voff = rowoff[j];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + t * 1024), 16, voff, cb4, 0, 0);
                }
            }
        } else {
            float* As = reinterpret_cast<float*>(reinterpret_cast<char*>(a_buf) + img);
            for (int e = tid; e < n * KS; e += NT) {
                const int r = e / KS, c = e % KS, col = k0 + c;
                float v = 0.f;
                if (col < p.c_in) {
                    const int64_t row = in_list[list_base + r];
                    v = (col < p.c_in_a) ? p.in_a[row * p.c_in_a + col] : p.in_b[row * p.c_in_b + (col - p.c_in_a)];
                }
                As[r * KS + 4 * ((c >> 2) ^ swz(r)) + (c & 3)] = v;
            }
        }
    };

    // float offset of this lane's chunk (channels 16 j + 4 lq .. +3 of row li) for the 16-channel group j
    int foff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) foff[j] = li * KS + 4 * ((4 * j + lq) ^ swz(li));

    constexpr int IMG = AF * 4;                           // bytes per A image
    int img = 0;                                          // byte offset of the image being multiplied
    f32x4 wc[NJ], wnx[NJ];                                // W fragments: current stage / next stage

    // One work item (offset, chunk) with NB active row blocks for this wave: all its slabs, straight-line
    // per stage (the only branches are the slab loop and the barrier).  MFMA step (j, e) takes
    // k = 16 j + 4 (lane >> 4) + e from both operands; the NB accumulators are interleaved, so
    // dependent MFMAs are NB x 32 cycles apart.  The accumulators live only inside the item: after its
    // last slab they are added into the LDS tile through the pair list's local output row -- batched
    // and branch-free: NB list words, then 4 NB tile reads, then 4 NB writes (3 LDS round trips); rows
    // beyond n go to the dummy row behind the tile.
    auto run_item = [&](auto nb_tag, const Item& item, const Item& next, bool has_next) {
        constexpr int NB = decltype(nb_tag)::value;
        f32x4 acc[NB > 0 ? NB : 1];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // MFMA groups [J0, J1) of the stage (a group = 16 channels = 4 MFMA steps per row block), fragments read from the image
        auto mma_part = [&](auto j0_tag, auto j1_tag) {
            constexpr int J0 = decltype(j0_tag)::value, J1 = decltype(j1_tag)::value;
            if constexpr (NB > 0 && J1 > J0) {
                const float* As = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a_buf) + img);
                f32x4 a[NB > 0 ? NB : 1][J1 > J0 ? J1 - J0 : 1];
#pragma unroll
                for (int j = J0; j < J1; ++j)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        a[b][j - J0] = *reinterpret_cast<const f32x4*>(As + (wm + WM * b) * (16 * KS) + foff[j]);
#pragma unroll
                for (int j = J0; j < J1; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[j][e], a[b][j - J0][e], acc[b], 0, 0, 0);
                // Fragment reads run one 16-channel group ahead of the MFMAs: left to itself the scheduler issues
                // the reads of group j+1 only after the last MFMA of group j, and the MFMA pipe then idles for an LDS
                // round trip four times per stage (both waves of a SIMD leave the barrier in lockstep, so neither
                // fills the other's gap).
                __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
#pragma unroll
                for (int j = J0; j + 1 < J1; ++j)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB, 0);
            }
        };
        auto mma = [&]() { mma_part(ic<0>{}, ic<NJ>{}); };
        auto stage_end = [&]() {
#ifdef LIDIFF_CONV_PROBE
            STAMP(tb0);
            if (!PROBE(4)) __syncthreads();
            t_barrier += __builtin_readcyclecounter() - tb0;
#else
            __syncthreads();                  // next stage landed (vmcnt) and visible; this image is free again
#endif
#pragma unroll
            for (int h = 0; h < NJ; ++h) wc[h] = wnx[h];
            img ^= IMG;
        };
#ifdef LIDIFF_CONV_PROBE
#define PHASE(acc, from) { const long long now_ = __builtin_readcyclecounter(); acc += now_ - from; from = now_; }
        long long tp = __builtin_readcyclecounter();
#else
#define PHASE(acc, from)
#endif
        for (int s = 0; s + 1 < nslab; ++s) {             // not the last slab: the next stage is the same item
            issue(img ^ IMG, item.k, s + 1, item.n, wnx);
            PHASE(t_issue, tp);
            mma();
            PHASE(t_mma, tp);
            stage_end();
#ifdef LIDIFF_CONV_PROBE
            tp = __builtin_readcyclecounter();
#endif
        }
        if (has_next) {                                   // last slab: the next stage opens the next item
            load_rows(next);
            issue(img ^ IMG, next.k, 0, next.n, wnx);
        }
        PHASE(t_issue, tp);
        mma();
        PHASE(t_mma, tp);
        STAMP(tf0);
        if constexpr (NB > 0) {
            if (!PROBE(16)) {
                // per element: its list word (one b128 read per block), one address add, read, add, write.
                // Round 3 tried two other forms, both measured and dropped (profiles/r03_dense_ablation.txt):
                //  * the add inside the MFMA -- accumulators initialised from the tile rows, written back after the last slab, no
                //    VALU arithmetic: +2 % on 256 -> 256, +4.6 % on 128 -> 128 at stride 8, but one 27 x C_in-term fp32 chain per
                //    output instead of 27 short ones: 10x the rounding error against the float64 oracle (5.1e-5 vs 4.7e-6);
                //  * list words, addresses and tile reads issued in FRONT of the last slab's MFMAs (they do not depend on the
                //    results): 252 VGPRs, 4-7 % SLOWER (98 -> 94 TFLOP/s on 256 -> 256, 84 -> 78 on 128 -> 128).
                // The MFMAs run with the operands SWAPPED (W fragment as A, the gathered rows as B): the transposed product
                // leaves in each lane four CONSECUTIVE channels (16 wn + 4 lq .. + 3) of ONE pair row (16 b + li) -- one 16-byte
                // read-modify-write of the tile per block instead of four 4-byte ones to four different rows, one list word per
                // block instead of four (round 4; the sums and their order are the same: bit-identical results).
                const OutT* ol = out_list + item.k * BM + item.start + li;
                char* accb = reinterpret_cast<char*>(acc_lds);
                int addr[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) addr[b] = tile_addr((int)ol[16 * (wm + WM * b)], 4 * wn + lq);
                f32x4 old[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) old[b] = *reinterpret_cast<const f32x4*>(accb + addr[b]);
#pragma unroll
                for (int b = 0; b < NB; ++b) *reinterpret_cast<f32x4*>(accb + addr[b]) = old[b] + acc[b];
            }
        }
#ifdef LIDIFF_CONV_PROBE
        t_flush += __builtin_readcyclecounter() - tf0;
#endif
        stage_end();
    };

    // ---- packed mode: a stage = SEG segments x SEGR rows, every segment its own offset ---------------
    auto run_packed = [&]() {
        if constexpr (SEG > 1) {
            constexpr int NSW = RBW < SEG ? RBW : SEG;        // distinct segments among this wave's row blocks
            constexpr int BLQ = RBW / NSW;                    // consecutive local blocks sharing a segment
            struct Desc { int k[NSW], start[NSW], n[NSW]; };
            const int npack = (nwork + SEG - 1) / SEG;
            auto load_pack = [&](int pack, Desc& d) {
#pragma unroll
                for (int q = 0; q < NSW; ++q) {
                    const int sid = (wm + WM * (q * BLQ)) / BPS;
                    const int wi = pack * SEG + sid;
                    const int e = __builtin_amdgcn_readfirstlane(work[min(wi, Cfg::WORK_INTS - 1)]);
                    const bool ok = wi < nwork;
                    d.k[q] = ok ? (e & 0xff) : 0;
                    d.start[q] = ok ? ((e >> 8) & 0xff) * SEGR : 0;
                    d.n[q] = ok ? (e >> 16) : 0;
                }
#pragma unroll
                for (int j = 0; j < T; ++j) {                 // this lane's gather rows: its image row's segment
                    const int r = RPI * (wave + NW * j) + lane / NCHK;
                    const int wi = pack * SEG + r / SEGR, idx = r % SEGR;
                    const int e = wi < nwork ? work[wi] : 0;
                    const bool ok = (T * NW == NINST || wave + NW * j < NINST) && idx < (e >> 16);
                    rowv[j] = ok ? in_list[(e & 0xff) * BM + ((e >> 8) & 0xff) * SEGR + idx] : -1;
                }
                set_rowoff(p.c_in_a * 4);
            };
            f32x4 wcs[NSW][NJ], wns[NSW][NJ];
            auto issue_packed = [&](int img_, const Desc& d, int slab, f32x4 (*w)[NJ]) {
                __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.wp), 0, (int)((size_t)p.k_vol * nslab32 * 32 * p.c_out * 4), 0x00020000);
#pragma unroll
                for (int q = 0; q < NSW; ++q) {
                    const int ws = (d.k[q] * nslab32 + slab) * w_slab_bytes;
                    w[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_lane_off, ws, 0));
                    w[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_lane_off + 1024, ws, 0));
                }
                const int k0 = slab * KS;
                const bool from_a = k0 < p.c_in_a;
                const float* src = from_a ? p.in_a : p.in_b;
                const int cw4 = (from_a ? p.c_in_a : p.c_in_b) * 4;
                const int cb4 = (from_a ? k0 : k0 - p.c_in_a) * 4;
                if (cw4 != rows_cw4) set_rowoff(cw4);
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(src), 0, (int)(p.m_in * cw4), 0x00020000);
                char* dst = reinterpret_cast<char*>(a_buf) + img_;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const int t = wave + NW * j;
                    if (T * NW == NINST || t < NINST) {
                        const int voff = rowoff[j];
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + t * 1024), 16, voff, cb4, 0, 0);
                    }
                }
            };
            Desc dc, dn;
            load_pack(0, dc);
            dn = dc;
            if (npack > 0) issue_packed(0, dc, 0, wcs);
            __syncthreads();
            for (int pack = 0; pack < npack; ++pack) {
                f32x4 acc[RBW];
#pragma unroll
                for (int b = 0; b < RBW; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                auto mma = [&]() {
                    const float* As = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a_buf) + img);
                    f32x4 a[RBW][NJ];
#pragma unroll
                    for (int b = 0; b < RBW; ++b)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            a[b][j] = *reinterpret_cast<const f32x4*>(As + (wm + WM * b) * (16 * KS) + foff[j]);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int b = 0; b < RBW; ++b)
                                acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wcs[b / BLQ][j][e], a[b][j][e], acc[b], 0, 0, 0);
                };
                auto stage_end = [&]() {
#ifdef LIDIFF_CONV_PROBE
                    STAMP(tb0);
                    __syncthreads();
                    t_barrier += __builtin_readcyclecounter() - tb0;
#else
                    __syncthreads();
#endif
#pragma unroll
                    for (int q = 0; q < NSW; ++q)
#pragma unroll
                        for (int h = 0; h < NJ; ++h) wcs[q][h] = wns[q][h];
                    img ^= IMG;
                };
#ifdef LIDIFF_CONV_PROBE
                long long tp = __builtin_readcyclecounter();
#endif
                for (int sl = 0; sl + 1 < nslab; ++sl) {
                    issue_packed(img ^ IMG, dc, sl + 1, wns);
                    PHASE(t_issue, tp);
                    mma();
                    PHASE(t_mma, tp);
                    stage_end();
#ifdef LIDIFF_CONV_PROBE
                    tp = __builtin_readcyclecounter();
#endif
                }
                if (pack + 1 < npack) {
                    load_pack(pack + 1, dn);
                    issue_packed(img ^ IMG, dn, 0, wns);
                }
                PHASE(t_issue, tp);
                mma();
                PHASE(t_mma, tp);
                STAMP(tf0);
                // flush: every block through its own segment's pair list.  Unlike a single-offset stage, the
                // segments of a pack can hit the SAME output row (one row, several offsets), so the adds are
                // ordered: the wm groups take turns (barrier in between), and inside a wave the segments go one
                // after the other (LDS executes a wave's accesses in order).  Fixed order => deterministic.
                // (Round 4 tried ordering by OFFSET instead -- the segments of one offset touch every output row at most once and
                // can be flushed by all waves at once, runs of equal offsets separated by barriers: 3-4 runs per pack of 8
                // segments, and each needs a workgroup barrier: 388 vs 338 us on 64 -> 64 at stride 4; not kept.)
                // The list words and tile addresses of ALL of the wave's segments first (independent of each other and of the
                // order); only the read-modify-writes of the tile are ordered, segment after segment.
                int addr[NSW][BLQ];
#pragma unroll
                for (int q = 0; q < NSW; ++q)
#pragma unroll
                    for (int g = 0; g < BLQ; ++g) {
                        const int gb = wm + WM * (q * BLQ + g);
                        addr[q][g] = tile_addr((int)out_list[dc.k[q] * BM + dc.start[q] + 16 * (gb % BPS) + li], 4 * wn + lq);
                    }
                for (int round = 0; round < WM; ++round) {
                    if (wm == round) {
                        char* accb = reinterpret_cast<char*>(acc_lds);
#pragma unroll
                        for (int q = 0; q < NSW; ++q) {
                            if (dc.n[q] == 0) continue;          // no segment here: nothing to add
                            f32x4 old[BLQ];
#pragma unroll
                            for (int g = 0; g < BLQ; ++g) old[g] = *reinterpret_cast<const f32x4*>(accb + addr[q][g]);
#pragma unroll
                            for (int g = 0; g < BLQ; ++g) *reinterpret_cast<f32x4*>(accb + addr[q][g]) = old[g] + acc[q * BLQ + g];
                            asm volatile("" ::: "memory");       // keep the segments' read-modify-writes in order
                        }
                    }
                    if (WM > 1) __syncthreads();
                }
#ifdef LIDIFF_CONV_PROBE
                t_flush += __builtin_readcyclecounter() - tf0;
#endif
                stage_end();
                dc = dn;
            }
        }
    };

    // ---- main loop: work items, loads one stage ahead ------------------------------------------
    STAMP(t_loop);
    if (packed) {
        run_packed();
    } else {
    Item item = load_item(0);
    load_rows(item);
    if (nwork > 0) issue(0, item.k, 0, item.n, wc);
    __syncthreads();
    for (int wi = 0; wi < nwork; ++wi) {
        const Item next = load_item(wi + 1);
        const bool has_next = wi + 1 < nwork;
        const int nrb = (item.n + 15) >> 4;
        const int nb = min(RBW, max(0, (nrb - wm + WM - 1) / WM));
        switch (nb) {
            case 0: run_item(ic<0>{}, item, next, has_next); break;
            case 1: run_item(ic<1>{}, item, next, has_next); break;
            case 2: if constexpr (RBW >= 2) run_item(ic<2>{}, item, next, has_next); break;
            case 3: if constexpr (RBW >= 4) run_item(ic<3>{}, item, next, has_next); break;
            case 4: if constexpr (RBW >= 4) run_item(ic<4>{}, item, next, has_next); break;
            case 5: if constexpr (RBW >= 8) run_item(ic<5>{}, item, next, has_next); break;
            case 6: if constexpr (RBW >= 8) run_item(ic<6>{}, item, next, has_next); break;
            case 7: if constexpr (RBW >= 8) run_item(ic<7>{}, item, next, has_next); break;
            default: if constexpr (RBW >= 8) run_item(ic<8>{}, item, next, has_next); break;
        }
        item = next;
    }
    }

    STAMP(t_epi);
    // ---- epilogue: BN scale/shift, residual, ReLU; one coalesced float4 store per 4 channels --
    // A thread keeps its channel quad over the whole loop when NT is a multiple of BN / 4 (every instantiation but the 3 x 2 /
    // 6 x 1 wave grids of 96 / 48 columns): scale and shift are loaded once; the residual rows of a batch of EB elements are all
    // requested before the first is used (the stores to `out` may alias the epilogue's inputs as far as the compiler knows, so
    // the plain loop waited for one global latency per element).
    {
        constexpr int CQ = BN / 4;
        constexpr bool FIXED_CQ = NT % CQ == 0;
        constexpr int EB = 4;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FIXED_CQ) {
            if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + n0 + 4 * (tid % CQ));
            if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + n0 + 4 * (tid % CQ));
        }
        const int total = rows_here * CQ;
        for (int e0 = tid; e0 < total; e0 += EB * NT) {
            float4 res[EB];
            int64_t oo[EB];
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int e = e0 + i * NT;
                const int r = min(e, total - 1) / CQ, cq = min(e, total - 1) % CQ;
                oo[i] = (int64_t)orow[r] * p.c_out + n0 + 4 * cq;
                res[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.residual && e < total) res[i] = *reinterpret_cast<const float4*>(p.residual + oo[i]);
            }
#pragma unroll
            for (int i = 0; i < EB; ++i) {
                const int e = e0 + i * NT;
                if (e >= total) break;
                const int r = e / CQ, cq = e % CQ;
                const int col = n0 + 4 * cq;
                float4 v = reinterpret_cast<const float4*>(acc_lds)[r * CQ + (cq ^ (r & (SWZ - 1)))];
                if (p.tail) {                          // contributions computed elsewhere (the non-centre offsets), fixed order
                    const int orw = orow[r];
                    // (qe clamped to the tail's rows: a no-op unless a bounded tail map overflowed -- then the step is redone)
                    for (int q = p.tail_ptr[orw], qe = min(p.tail_ptr[orw + 1], (int)p.tail_rows); q < qe; ++q) {
                        const float4 s4 = *reinterpret_cast<const float4*>(p.tail + (int64_t)p.tail_idx[q] * p.c_out + col);
                        v.x += s4.x; v.y += s4.y; v.z += s4.z; v.w += s4.w;
                    }
                }
                if constexpr (!FIXED_CQ) {
                    sc = p.scale ? *reinterpret_cast<const float4*>(p.scale + col) : make_float4(1.f, 1.f, 1.f, 1.f);
                    sh = p.shift ? *reinterpret_cast<const float4*>(p.shift + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (p.scale) { v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
                if (p.shift) { v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w; }
                if (p.residual) { v.x += res[i].x; v.y += res[i].y; v.z += res[i].z; v.w += res[i].w; }
                if (p.relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<float4*>(p.out + oo[i]) = v;
            }
        }
    }
#ifdef LIDIFF_CONV_PROBE
    if (p.timeline != nullptr && lane == 0 && (wave == 0 || wave == NW / 2)) {   // the two waves of SIMD 0
        STAMP(t_end);
        long long* d = p.timeline + ((int64_t)bid * 2 + (wave != 0)) * 10;
        d[0] = t_loop - t_start; d[1] = t_epi - t_loop; d[2] = t_end - t_epi; d[3] = t_barrier; d[4] = t_flush;
        d[5] = packed ? (nwork + SEG - 1) / SEG : nwork; d[6] = nslab; d[7] = __builtin_amdgcn_s_memrealtime() - rt_start; d[8] = t_issue; d[9] = t_mma;
    }
#endif
}

template <int BM, int WN, int WM, int KS, bool VEC>
__global__ __launch_bounds__(64 * WN * WM) void spconv_fwd_kernel(const ConvParams p_launch) {
    conv_tile<BM, WN, WM, KS, VEC>(p_launch, blockIdx.x);
}

// W [K, c_in, c_out] row-major  ->  [K][slab32][c_out/16][j 0..1][lane 0..63][e 0..3]  with
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

template <int BM, int WN, int WM, int KS, bool VEC>
static int launch_fwd(const ConvParams& p, hipStream_t st) {
    using Cfg = ConvCfg<BM, WN, WM, KS>;
    const size_t lds = Cfg::lds_bytes(p.k_vol);
    LIDIFF_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded");
    auto kern = spconv_fwd_kernel<BM, WN, WM, KS, VEC>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / Cfg::BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

template <int BM, int WN, int WM>
static int dispatch_fwd(const ConvParams& p, bool vec, hipStream_t st) {
    if (!vec) return launch_fwd<BM, WN, WM, 32, false>(p, st);
    // 64-channel stages halve the per-stage costs of dense maps; low-density maps (hint from the caller)
    // take the 32-channel kernel, whose tiles can pack several offsets into one stage
    if (p.c_in_a % 64 == 0 && p.c_in_b % 64 == 0 && !(p.flags & LIDIFF_CONV_SPARSE_MAP))
        return launch_fwd<BM, WN, WM, 64, true>(p, st);
    return launch_fwd<BM, WN, WM, 32, true>(p, st);
}

}  // namespace lidiff

using namespace lidiff;

static int g_conv_probe = 0;
static long long* g_conv_timeline = nullptr;
#ifdef LIDIFF_CONV_PROBE
extern "C" void lidiff_debug_set_conv_probe(int flags) { g_conv_probe = flags; }
extern "C" void lidiff_debug_set_conv_timeline(long long* buf) { g_conv_timeline = buf; }
#endif

extern "C" int64_t lidiff_spconv_packed_weight_floats(int32_t k_vol, int32_t c_in, int32_t c_out) {
    return (int64_t)k_vol * ((c_in + 31) / 32) * 32 * c_out;
}

extern "C" int lidiff_spconv_pack_weights(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out,
                                          float* w_packed, void* stream) {
    LIDIFF_CHECK_ARG(w != nullptr && w_packed != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && c_in > 0, "kernel volume must be 1..27, c_in > 0");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    const int nslab = (c_in + 31) / 32;
    const int64_t total = lidiff_spconv_packed_weight_floats(k_vol, c_in, c_out);
    pack_weights_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w, k_vol, c_in, c_out, nslab,
                                                                                         w_packed, total);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t lidiff_spconv_fwd_kernel_id(int32_t c_in_a, int32_t c_in_b, int32_t c_out, int32_t k_vol, int32_t has_nbr,
                                               int32_t has_row_order, int32_t flags) {
    static const int32_t some = 0;                       // any non-null address: only nullness is looked at
    ConvParams p{};
    p.nbr = has_nbr ? &some : nullptr; p.row_order = has_row_order ? &some : nullptr;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out; p.k_vol = k_vol; p.flags = flags;
    p.m_in = p.m_out = 1;
    if (!(flags & LIDIFF_CONV_TILE_ONLY) && rows_kernel_applies(p)) return 2;
    if (!(flags & LIDIFF_CONV_TILE_ONLY) && thin_kernel_applies(p)) return 3;
    return 0;
}

extern "C" int lidiff_spconv_fwd(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                 const float* w_packed, const int32_t* nbr, int32_t k_vol, int64_t m_in,
                                 int64_t m_out, int32_t c_out, float* out, const float* ep_scale,
                                 const float* ep_shift, const float* residual, int32_t relu,
                                 const int32_t* row_order, int32_t replicas, int32_t flags, const float* tail,
                                 const int32_t* tail_ptr, const int32_t* tail_idx, int64_t tail_rows,
                                 const int32_t* d_m_out, void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0, "in_a / c_in_a");
    LIDIFF_CHECK_ARG((tail == nullptr) == (tail_ptr == nullptr) && (tail == nullptr) == (tail_idx == nullptr),
                     "tail, tail_ptr and tail_idx must be all set or all null");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(nbr != nullptr || (k_vol == 1 && m_in == m_out), "identity map needs K=1, m_in==m_out");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    LIDIFF_CHECK_ARG(m_out >= 0 && m_in >= 0, "negative rows");
    LIDIFF_CHECK_ARG(replicas >= 1, "replicas must be >= 1");
    if (m_out == 0) return 0;
    LIDIFF_CHECK_ARG(m_in > 0, "outputs without inputs");
    ConvParams p{};
    p.in_a = in_a; p.in_b = in_b; p.wp = w_packed; p.nbr = nbr; p.row_order = row_order; p.out = out;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.tail = tail; p.tail_ptr = tail_ptr; p.tail_idx = tail_idx; p.tail_rows = tail_rows;
    p.m_in = m_in; p.m_out = m_out; p.d_m_out = d_m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu; p.flags = flags; p.replicas = replicas; p.probe = g_conv_probe; p.timeline = g_conv_timeline;
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(w_packed) && al16(out) && al16(ep_scale) && al16(ep_shift) && al16(residual) && al16(tail),
                     "w_packed/out/epilogue/tail pointers must be 16-byte aligned");
    const bool fits32 = m_in * (int64_t)c_in_a * 4 < (1ll << 31) && m_in * (int64_t)c_in_b * 4 < (1ll << 31);
    LIDIFF_CHECK_ARG(fits32, "a feature matrix exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(c_in_b == 0 || c_in_a % 32 == 0, "with two inputs c_in_a must be a multiple of 32 (slab size)");
    const bool vec = c_in_a % 32 == 0 && c_in_b % 32 == 0 && al16(in_a) && al16(in_b);   // else: scalar gather
    hipStream_t st = (hipStream_t)stream;
    // identity maps: consecutive rows, one offset -- the streaming row GEMM of spconv_rows.hip (bit-identical results)
    if (!(flags & LIDIFF_CONV_TILE_ONLY) && al16(in_a) && al16(in_b) && rows_kernel_applies(p)) return launch_fwd_rows(p, st);
    // inputs of <= 4 channels (the stems): nothing to multiply, a VALU kernel over the table
    if (!(flags & LIDIFF_CONV_TILE_ONLY) && thin_kernel_applies(p)) return launch_fwd_thin(p, st);
    if (c_out % 128 == 0) return dispatch_fwd<128, 8, 1>(p, vec, st);
    if (c_out % 96 == 0) {
        // low-density maps: two 48-column tiles of 3 x 2 waves -- their packed stages hold 8 offsets (own W
        // registers per 16-row segment) instead of the 4 a 6 x 1 wave grid has registers for
        if (vec && (flags & LIDIFF_CONV_SPARSE_MAP)) return dispatch_fwd<128, 3, 2>(p, vec, st);
        return dispatch_fwd<128, 6, 1>(p, vec, st);
    }
    if (c_out % 64 == 0) {
        // 64-column kernel_size-3 layers (stride-2/4 levels: 1.5-10 neighbours per voxel): 256-row tiles -- the same 64 KB of
        // accumulators as a 128 x 128 tile, twice the pairs per offset and stage for the same per-stage costs: 5-25 % faster
        // on the bench maps (profiles/r03_tile256_probe.txt); the stride-2 "down" maps (8 offsets) lose 13-19 % and keep 128
        // rows, as do maps too small to fill the chip with 256-row tiles (LIDIFF_CONV_TILE_128: always 128 rows)
        if (vec && k_vol == 27 && !(flags & LIDIFF_CONV_TILE_128) && m_out * replicas >= 256 * 512)
            return launch_fwd<256, 4, 2, 32, true>(p, st);
        return dispatch_fwd<128, 4, 2>(p, vec, st);
    }
    if (c_out % 32 == 0) {
        // 32 columns: the other way round -- kernel_size 3 loses 20-35 % on 256-row tiles, the stride-2 maps gain 20 %
        if (vec && k_vol == 8 && !(flags & LIDIFF_CONV_TILE_128) && m_out * replicas >= 256 * 512)
            return launch_fwd<256, 2, 4, 32, true>(p, st);
        return dispatch_fwd<128, 2, 4>(p, vec, st);
    }
    return dispatch_fwd<128, 1, 8>(p, vec, st);
}

namespace lidiff {

// =======================================================================================
// Weight gradient of the sparse convolution (training path, models.py:180-217; ME: ConvolutionBackwardGPU):
//     dW[k][ci][co] = sum over the pairs (i, o) of offset k of  in[i][ci] * g[o][co]
// i.e. per offset a [c_in x P_k] x [P_k x c_out] product whose K dimension is the offset's pair list (the
// ME-layout rulebook: pairs sorted by offset).  One workgroup of 8 waves owns a [16 NBI ci x 128 CB co] tile of
// dW[k] -- up to the whole 256 x 256 block, so that every pair row is gathered once -- entirely in MFMA
// accumulators (wave w: co blocks w CB .. w CB + CB - 1, all NBI ci blocks), for one slice of the offset's pairs.
// Per chunk of 64 pairs the gathered rows of `in` and `g` are stored row-major in LDS with a pitch = 16 mod 32
// words: the v_mfma_f32_16x16x4_f32 operand of a lane is then in[pair 4 s + (lane >> 4)][ci block + (lane & 15)],
// a conflict-free 4-byte read, and nothing is transposed.  The next chunk's rows are prefetched into registers
// while the current one is multiplied.  Slices are summed with fp32 atomics (as ME does), so dW is deterministic
// only up to the order of those adds.

template <int NBI, int CB, bool IDENT>
__global__ __launch_bounds__(512) void spconv_bwd_w_kernel(const float* __restrict__ in_a, int c_in_a,
                                                           const float* __restrict__ in_b, int c_in_b,
                                                           const float* __restrict__ g, const int32_t* __restrict__ pairs_in,
                                                           const int32_t* __restrict__ pairs_out,
                                                           const int32_t* __restrict__ offset_ptr, int64_t m_out,
                                                           int c_out, int64_t per, float* __restrict__ dw,
                                                           float* __restrict__ part, int k_vol) {
    constexpr int CIT = 16 * NBI, COT = 128 * CB;        // tile extents
    constexpr int PA = CIT + 16, PG = COT + 16;          // LDS row pitches (words): = 16 mod 32
    constexpr int A4 = kDwPairs * (CIT / 4), G4 = kDwPairs * (COT / 4);   // float4 pieces per chunk
    constexpr int NA = (A4 + 511) / 512, NG = (G4 + 511) / 512;          // ... per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a_s = reinterpret_cast<float*>(smem);          // [pair][ci]
    float* g_s = a_s + kDwPairs * PA;                     // [pair][co]
    const int c_in = c_in_a + c_in_b;
    // offsets vary fastest over the grid: workgroups in flight together work on the same stretch of rows under
    // different offsets, i.e. on the same rows of g and neighbouring rows of in (L2 reuse)
    const int co_tiles = (c_out + COT - 1) / COT;
    const int ci0 = (blockIdx.y / co_tiles) * CIT, co0 = (blockIdx.y % co_tiles) * COT;
    // this workgroup's slot: `per` pairs (whole chunks) of the offset the slot belongs to -- slots go to the offsets in proportion
    // to their pair counts (spconv.h), offsets varying fastest over consecutive slots
    const int slot = (int)(blockIdx.z * gridDim.x + blockIdx.x);
    int k, local;
    int64_t p_lo, p_hi;
    if (!dw_slot_offset(IDENT ? nullptr : offset_ptr, k_vol, m_out, per, slot, k, local, p_lo, p_hi)) return;       // idle slot
    const int64_t s_lo = p_lo + (int64_t)local * per, s_hi = min(p_hi, s_lo + per);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;

    float4 pa[NA], pg[NG];                                // the next chunk's rows, in flight
    unsigned keep = 0;                                    // which of them are real (bit t: pa[t], bit 16 + t: pg[t])
    // Branch-free and in two rounds -- all pair indices, then all rows -- so that the loads of a chunk are in
    // flight together (a per-piece "if valid: index, then row" compiles to a chain of dependent round trips).
    // Pieces outside the slice or the channel range load a clamped address and are zeroed when stashed.
    auto fetch = [&](int64_t base) {
        int ra[NA], rg[NG];                               // IDENT: the identity map of a kernel_size-1 convolution
        keep = 0;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int64_t pp = min(base + min(tid + 512 * t, A4 - 1) / (CIT / 4), s_hi - 1);
            ra[t] = IDENT ? (int)pp : pairs_in[pp];
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int64_t pp = min(base + min(tid + 512 * t, G4 - 1) / (COT / 4), s_hi - 1);
            rg[t] = IDENT ? (int)pp : pairs_out[pp];
        }
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int e = tid + 512 * t, ec = min(e, A4 - 1), pr = ec / (CIT / 4), ci = ci0 + (ec % (CIT / 4)) * 4;
            const int cic = min(ci, c_in - 4);
            const float* src = cic < c_in_a ? in_a + (unsigned)(ra[t] * c_in_a + cic)
                                            : in_b + (unsigned)(ra[t] * c_in_b + (cic - c_in_a));
            pa[t] = *reinterpret_cast<const float4*>(src);
            keep |= (e < A4 && base + pr < s_hi && ci < c_in) ? 1u << t : 0u;
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int e = tid + 512 * t, ec = min(e, G4 - 1), pr = ec / (COT / 4), co = co0 + (ec % (COT / 4)) * 4;
            pg[t] = *reinterpret_cast<const float4*>(g + (unsigned)(rg[t] * c_out + min(co, c_out - 4)));
            keep |= (e < G4 && base + pr < s_hi && co < c_out) ? 1u << (16 + t) : 0u;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int e = tid + 512 * t;
            const float m = (keep >> t) & 1u ? 1.f : 0.f;
            if (e < A4)
                *reinterpret_cast<float4*>(a_s + (e / (CIT / 4)) * PA + (e % (CIT / 4)) * 4) =
                    make_float4(pa[t].x * m, pa[t].y * m, pa[t].z * m, pa[t].w * m);
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int e = tid + 512 * t;
            const float m = (keep >> (16 + t)) & 1u ? 1.f : 0.f;
            if (e < G4)
                *reinterpret_cast<float4*>(g_s + (e / (COT / 4)) * PG + (e % (COT / 4)) * 4) =
                    make_float4(pg[t].x * m, pg[t].y * m, pg[t].z * m, pg[t].w * m);
        }
    };

    f32x4 acc[NBI][CB];
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s_lo < s_hi) fetch(s_lo);
    for (int64_t base = s_lo; base < s_hi; base += kDwPairs) {
        __syncthreads();                                  // the previous chunk has been multiplied
        stash();
        __syncthreads();
        if (base + kDwPairs < s_hi) fetch(base + kDwPairs);
        const int steps = (int)min((int64_t)kDwPairs / 4, (s_hi - base + 3) / 4);
        for (int s4 = 0; s4 < steps; ++s4) {              // MFMA step: pairs 4 s4 .. 4 s4 + 3 (k = lane >> 4)
            const float* ar = a_s + (4 * s4 + lq) * PA + li;
            const float* gr = g_s + (4 * s4 + lq) * PG + 16 * CB * wave + li;
            float bf[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) bf[c] = gr[16 * c];
#pragma unroll
            for (int b = 0; b < NBI; ++b) {
                const float af = ar[16 * b];
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[c], acc[b][c], 0, 0, 0);
            }
        }
    }
    // D layout: col = lane & 15 (co), row = 4 (lane >> 4) + reg (ci).  With a workspace every pair slice stores its partial
    // tile (summed in slice order by dw_reduce_kernel: deterministic); without one the slices meet in dw by fp32 atomics.
    float* dwk = part ? part + (int64_t)slot * c_in * c_out : dw + (int64_t)k * c_in * c_out;
#pragma unroll
    for (int b = 0; b < NBI; ++b)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + 16 * b + 4 * lq + r, co = co0 + 16 * (CB * wave + c) + li;
                if (ci < c_in && co < c_out) {
                    if (part) dwk[(int64_t)ci * c_out + co] = acc[b][c][r];
                    else if (acc[b][c][r] != 0.f) atomicAdd(dwk + (int64_t)ci * c_out + co, acc[b][c][r]);
                }
            }
}

// dw[k][e] = sum over the offset's slots, in slot order (an offset without pairs: zeros)
__global__ void dw_reduce_kernel(const float* __restrict__ part, int64_t n_k, int k_vol, const int32_t* __restrict__ offset_ptr,
                                 int64_t m_ident, int64_t per, float* __restrict__ dw) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (e >= n_k) return;
    int first = 0, count = 0;
    for (int kk = 0; kk <= k; ++kk) {
        const int64_t lo = offset_ptr ? offset_ptr[kk] : 0, hi = offset_ptr ? offset_ptr[kk + 1] : m_ident;
        first += count;
        count = (int)((hi - lo + per - 1) / per);
    }
    float v = 0.f;
    const float* src = part + (int64_t)first * n_k + e;
    int s = 0;
    for (; s + 8 <= count; s += 8) {                      // eight loads in flight, added in slot order
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = src[(int64_t)(s + i) * n_k];
#pragma unroll
        for (int i = 0; i < 8; ++i) v += t[i];
    }
    for (; s < count; ++s) v += src[(int64_t)s * n_k];
    dw[(int64_t)k * n_k + e] = v;
}

int64_t dw_pairs_per_slot(int64_t n_pairs, int64_t slices, int k_vol) {
    if (slices <= 1) return (int64_t)1 << 40;             // one slot per offset
    // sum_k ceil(P_k / per) <= n_pairs / per + k_vol <= slices * k_vol
    const int64_t slots = slices * k_vol - k_vol;
    return ceil_div(ceil_div(n_pairs > 0 ? n_pairs : 1, slots), (int64_t)kDwPairs) * kDwPairs;
}

int64_t dw_slices(int c_in, int c_out, int k_vol, int64_t n_pairs, int cit, int cot) {
    const int tiles = (int)(ceil_div(c_in, cit) * ceil_div(c_out, cot));
    // workgroups per launch.  With EQUAL slices per offset, whose pair counts differ widely (the centre offset
    // holds every row), many short slices balanced best: 2 048 (bf16 step 1 024: 104.7 ms, 2 048: 102.0, 4 096: 103.4, 8 192: 107.1).
    // With slots handed out in proportion to the pair counts (dw_pairs_per_slot) every workgroup has the same work, and fewer, longer
    // slots win -- less partial-tile traffic through the workspace: 2 048: 84.1 ms, 1 024: 80.7, 768: 80.2, 512: 79.8, 384: 80.4
    // (fp32: 164 / 160 / - / 162) -- but at least 4 chunks each
    constexpr int64_t target = 768;
    int64_t slices = ceil_div(target, (int64_t)tiles * k_vol);
    const int64_t max_slices = max((int64_t)1, n_pairs / k_vol / (4 * kDwPairs));
    return max((int64_t)1, min(slices, max_slices));
}

template <int NBI, int CB, bool IDENT>
static int launch_bwd_w(const float* in_a, int c_in_a, const float* in_b, int c_in_b, const float* g,
                        const int32_t* pin, const int32_t* pout, const int32_t* off, int k_vol, int64_t m_out,
                        int64_t n_pairs, int c_out, float* dw, float* workspace, hipStream_t st) {
    constexpr int CIT = 16 * NBI, COT = 128 * CB;
    const size_t lds = (size_t)kDwPairs * ((CIT + 16) + (COT + 16)) * 4;
    auto kern = spconv_bwd_w_kernel<NBI, CB, IDENT>;
    static thread_local bool configured = false;
    if (!configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    const int c_in = c_in_a + c_in_b;
    const int tiles = (int)(ceil_div(c_in, CIT) * ceil_div(c_out, COT));
    const int64_t slices = dw_slices(c_in, c_out, k_vol, n_pairs, CIT, COT);
    const int64_t per = dw_pairs_per_slot(n_pairs, slices, k_vol);
    const int64_t n_k = (int64_t)c_in * c_out;
    float* part = (workspace != nullptr && slices > 1) ? workspace : nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)k_vol, (unsigned)tiles, (unsigned)slices), dim3(512), lds, st, in_a, c_in_a,
                       in_b, c_in_b, g, pin, pout, off, m_out, c_out, per, dw, part, k_vol);
    if (part) dw_reduce_kernel<<<dim3((unsigned)ceil_div(n_k, 256), (unsigned)k_vol), 256, 0, st>>>(part, n_k, k_vol, off, m_out, per, dw);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

}  // namespace lidiff

extern "C" int64_t lidiff_spconv_bwd_w_workspace_floats(int32_t c_in, int32_t c_out, int32_t k_vol, int64_t n_pairs) {
    const int nbi = c_in > 128 ? 16 : c_in > 64 ? 8 : c_in > 32 ? 4 : 2;
    const int64_t slices = dw_slices(c_in, c_out, k_vol, n_pairs > 0 ? n_pairs : 1, 16 * nbi, c_out > 128 ? 256 : 128);
    return slices > 1 ? slices * k_vol * (int64_t)c_in * c_out : 0;
}

extern "C" int lidiff_spconv_bwd_w(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                   const float* grad_out, const int32_t* pairs_in, const int32_t* pairs_out,
                                   const int32_t* offset_ptr, int64_t n_pairs, int32_t k_vol, int64_t m_in,
                                   int64_t m_out, int32_t c_out, float* dw, float* workspace, void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0 && grad_out != nullptr && dw != nullptr, "null pointer");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    const bool identity = pairs_in == nullptr && pairs_out == nullptr && offset_ptr == nullptr;
    LIDIFF_CHECK_ARG(identity ? (k_vol == 1 && m_in == m_out) : (pairs_in && pairs_out && offset_ptr),
                     "rulebook pointers must be all set, or all null for the identity map (K=1, m_in==m_out)");
    LIDIFF_CHECK_ARG(c_in_a % 4 == 0 && c_in_b % 4 == 0 && c_out % 4 == 0, "channel counts must be multiples of 4");
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(in_a) && al16(in_b) && al16(grad_out), "feature pointers must be 16-byte aligned");
    LIDIFF_CHECK_ARG(m_in * (int64_t)(c_in_a > c_in_b ? c_in_a : c_in_b) < (1ll << 31) && m_out * (int64_t)c_out < (1ll << 31),
                     "a feature matrix exceeds 2^31 elements (32-bit row offsets)");
    if (identity) n_pairs = m_out;
    if (m_out == 0 || n_pairs <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int c_in = c_in_a + c_in_b;
    const int nbi = c_in > 128 ? 16 : c_in > 64 ? 8 : c_in > 32 ? 4 : 2;      // ci blocks of the tile (<= 256 channels)
#define LIDIFF_DW(NBI, CB)                                                                                          \
    return identity ? launch_bwd_w<NBI, CB, true>(in_a, c_in_a, in_b, c_in_b, grad_out, pairs_in, pairs_out, offset_ptr,  \
                                                  k_vol, m_out, n_pairs, c_out, dw, workspace, st)                     \
                    : launch_bwd_w<NBI, CB, false>(in_a, c_in_a, in_b, c_in_b, grad_out, pairs_in, pairs_out, offset_ptr, \
                                                   k_vol, m_out, n_pairs, c_out, dw, workspace, st)
    if (c_out > 128) {
        if (nbi == 16) LIDIFF_DW(16, 2);
        if (nbi == 8) LIDIFF_DW(8, 2);
        if (nbi == 4) LIDIFF_DW(4, 2);
        LIDIFF_DW(2, 2);
    }
    if (nbi == 16) LIDIFF_DW(16, 1);
    if (nbi == 8) LIDIFF_DW(8, 1);
    if (nbi == 4) LIDIFF_DW(4, 1);
    LIDIFF_DW(2, 1);
#undef LIDIFF_DW
}
