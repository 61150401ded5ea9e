// Sparse convolution, dense-map kernel for gfx950: one wave per SIMD, software-pipelined across stages.
//
// Same decomposition as spconv.hip (output-stationary, pair-compacted, fp32 MFMA v_mfma_f32_16x16x4_f32; a
// workgroup owns 128 output rows x 128 output channels and keeps the accumulator tile in LDS), re-scheduled around
// what the counters of that kernel showed on the 256-channel stride-8 / stride-16 layers (profiles/r01_pmc_*,
// DESIGN.md section 4.2): the MFMA pipe idled a third of the time because (1) all eight waves issued their loads
// in one burst right after the stage barrier and could not start multiplying until the load queue had taken them,
// (2) the two waves of a SIMD serialise on fp32 MFMAs, so the second wave's loads and flush queued behind the
// first wave's MFMA stream, (3) the first fragment reads of a stage and the read-add-write flush were exposed.
//
// Here a workgroup is FOUR waves, one per SIMD, each owning 32 output columns (two 16-column MFMA blocks) of all
// row blocks of a stage, with the whole register file (up to 512 VGPRs) to itself:
//   * every A fragment (one ds_read_b128) feeds two MFMA column blocks: half the LDS reads per MFMA;
//   * stage = (offset, <=128 pairs, 32-channel slab).  The gathered A rows of stage s+3 are requested by LDS-DMA
//     into a ring of FOUR 16 KB images while stage s is multiplied, the W fragments of stage s+2 go straight to a
//     second register set after the last MFMA of stage s has been issued: requests have two full stages to land and
//     are spread over the stage instead of bunched behind the barrier;
//   * the stage barrier is a raw s_barrier behind a COUNTED s_waitcnt vmcnt(8) (the requests of the two younger
//     stages stay in flight across it; __syncthreads() would drain them);
//   * fragments run one 16-channel group ahead of the MFMAs ACROSS the stage boundary: the first group of stage
//     s+1 is read into registers while the second group of stage s multiplies, so no stage opens with an LDS
//     round trip;
//   * the flush of an offset's accumulators into the LDS tile is one ds_add_f32 per element (no read, no VALU add,
//     no write-back); each output row occurs at most once per offset and a wave owns its columns, offsets are
//     flushed in ascending order by the same wave: the sum order is fixed, results are deterministic.
// Everything else (pair lists by wave ballot, source-side XOR swizzle of the DMA'd image, packed weights, fused
// BatchNorm / residual / ReLU epilogue, XCD-aware tile mapping, replicas) is as in spconv.hip.
//
// Applies to: c_out % 128 == 0, both input widths multiples of 32 with an even number of 32-channel slabs in total
// (64, 128, 192, 256, 384 ...), 16-byte aligned inputs, no low-density hint.  Everything else runs spconv.hip.
#include "spconv.h"

namespace lidiff {

namespace dense {

constexpr int BM = 128, BN = 128, KS = 32, NW = 4, NT = 256;
constexpr int IMG = BM * KS * 4;                  // 16 KB per A image
constexpr int RING = 4;
constexpr int T = 4;                              // LDS-DMA wave-instructions per wave and stage (16 x 1 KB / 4 waves)
constexpr int kDummy = BM * BN;                   // float index of the dummy accumulator row

// LDS map (bytes)
constexpr int L_ACC = RING * IMG;                              // 65536: (BM + 1) x BN floats
constexpr int L_IN = L_ACC + (BM + 1) * BN * 4;                // in_list [27][128] int32
constexpr int L_OUT = L_IN + 27 * BM * 4;                      // out_list [27][128] uint16 (float index of the row)
constexpr int L_CNT = L_OUT + 27 * BM * 2;                     // cnt[32]
constexpr int L_ITEMS = L_CNT + 32 * 4;                        // items[32]: k | n << 8
constexpr int L_OROW = L_ITEMS + 32 * 4;                       // output row of every tile row
constexpr int L_TOTAL = L_OROW + BM * 4;                       // 153 088
static_assert(L_TOTAL <= 160 * 1024, "LDS");
static_assert(27 * BM * 4 <= RING * IMG, "raw neighbour block is staged in the A ring");

}  // namespace dense

__global__ __launch_bounds__(dense::NT) __attribute__((amdgpu_waves_per_eu(1, 1)))
void spconv_fwd_dense_kernel(const ConvParams p_launch) {
    using namespace dense;
    ConvParams p = p_launch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = reinterpret_cast<float*>(smem + L_ACC);
    int32_t* in_list = reinterpret_cast<int32_t*>(smem + L_IN);
    uint16_t* out_list = reinterpret_cast<uint16_t*>(smem + L_OUT);
    int32_t* cnt = reinterpret_cast<int32_t*>(smem + L_CNT);
    int32_t* items = reinterpret_cast<int32_t*>(smem + L_ITEMS);
    int32_t* orow = reinterpret_cast<int32_t*>(smem + L_OROW);

    // tile mapping as in spconv.hip: column tiles of a row tile share an XCD, row tiles round-robin over the XCDs
    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tiles_all = p.tiles_m * p.replicas;
    const int tmr = (g / p.tiles_n) * 8 + xcd;
    if (tmr >= tiles_all) return;
    const int rep = tmr / p.tiles_m, tm = tmr - rep * p.tiles_m;
    p.in_a += (int64_t)rep * p.m_in * p.c_in_a;
    if (p.in_b) p.in_b += (int64_t)rep * p.m_in * p.c_in_b;
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;

    // ---- pair lists (ordered compaction per offset), accumulator tile cleared ---------------------
    for (int e = tid; e < BM * BN / 4; e += NT) reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = tid; r < rows_here; r += NT) orow[r] = p.row_order ? p.row_order[row0 + r] : (int32_t)(row0 + r);
    if (p.nbr == nullptr) {                       // kernel_size 1: identity map
        for (int r = tid; r < BM; r += NT) {
            const int64_t gr = min(row0 + r, p.m_out - 1);
            in_list[r] = p.row_order ? p.row_order[gr] : (int32_t)gr;
            out_list[r] = (uint16_t)(r < rows_here ? r * BN : kDummy);
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        int32_t* raw = reinterpret_cast<int32_t*>(smem);      // the A ring is idle here
        for (int e = tid; e < p.k_vol * BM; e += NT) {
            const int k = e / BM, r = e % BM;
            raw[e] = r < rows_here ? p.nbr[(int64_t)k * p.m_out + row0 + r] : -1;
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
                    out_list[k * BM + q] = (uint16_t)(r * BN);
                }
                pos += __popcll(m);
            }
#pragma unroll
            for (int c = 0; c < BM; c += 64)
                if (c + lane >= pos) out_list[k * BM + c + lane] = (uint16_t)kDummy;
            if (lane == 0) cnt[k] = pos;
        }
    }
    __syncthreads();
    if (wave == 0) {                              // items: the offsets that have pairs, ascending
        const int c = (lane < p.k_vol && lane < 32) ? cnt[lane] : 0;
        const unsigned long long m = __ballot(c > 0);
        if (c > 0) items[popc_below(m)] = lane | (c << 8);
        if (lane == 0) cnt[31] = __popcll(m);
    }
    __syncthreads();
    const int n_items = __builtin_amdgcn_readfirstlane(cnt[31]);
    const int nslab = p.c_in / KS;                // even (dense_kernel_applies)
    const int nslab_a = p.c_in_a / KS;
    const int nt16 = p.c_out >> 4;
    const int w_slab_bytes = nt16 * 512 * 4;      // one 32-channel slab of one offset in the packed weights
    const int w_lane_off = (((n0 >> 4) + 2 * wave) * 512 + lane * 4) * 4;
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (int)((size_t)p.k_vol * nslab * 32 * p.c_out * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in_a), 0, (int)(p.m_in * p.c_in_a * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in_b ? p.in_b : p.in_a), 0, (int)(p.in_b ? p.m_in * p.c_in_b * 4 : 0), 0x00020000);

    auto item_at = [&](int i) {                   // (k, n) of item i; beyond the end: an empty item (dummy stages)
        const int w = __builtin_amdgcn_readfirstlane(items[min(i, n_items - 1)]);
        return i < n_items ? w : (w & 0xff);
    };

    // ---- A-gather cursor: runs three stages ahead of the multiplication -----------------------------
    // lane's 16-byte piece of DMA instruction j: image row r = 8 (wave + 4 j) + lane / 8, chunk (lane % 8) XOR
    // swizzle(r) of the slab (source-side swizzle: the ds_read_b128 fragment reads are bank-conflict free)
    int chb[T], rowv[T], rowoff[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int r = 8 * (wave + NW * j) + (lane >> 3);
        chb[j] = 16 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    int a_item = 0, a_slab = 0, a_slot = 0;       // next stage to request: item, slab, ring slot (bytes)
    auto a_rows = [&](int w) {                     // gather rows of item word w (source a pitch)
        const int k = w & 0xff, n = w >> 8;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int r = 8 * (wave + NW * j) + (lane >> 3);
            rowv[j] = r < n ? in_list[k * BM + r] : -1;
            rowoff[j] = rowv[j] >= 0 ? rowv[j] * (p.c_in_a * 4) + chb[j] : (int)0x80000000;      // OOB -> zero fill
        }
    };
    auto a_issue = [&]() {
        const bool from_a = a_slab < nslab_a;
        const int cb4 = (from_a ? a_slab : a_slab - nslab_a) * (KS * 4);
        char* dst = smem + a_slot;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            lds_ptr_t d = (lds_ptr_t)(dst + (wave + NW * j) * 1024);
            if (from_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, d, 16, rowoff[j], cb4, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, d, 16, rowoff[j], cb4, 0, 0);
        }
        a_slot = (a_slot + IMG) & (RING * IMG - 1);
        if (++a_slab == nslab) {                   // next item
            a_slab = 0;
            a_rows(item_at(++a_item));
        } else if (a_slab == nslab_a) {            // a -> b: the row pitch changes
#pragma unroll
            for (int j = 0; j < T; ++j) rowoff[j] = rowv[j] >= 0 ? rowv[j] * (p.c_in_b * 4) + chb[j] : (int)0x80000000;
        }
    };

    // ---- W cursor: two stages ahead, alternating register sets -------------------------------------
    f32x4 w[2][2][2];                              // [set][column block][16-channel group]
    int w_item = 0, w_slab = 0, w_k = 0;
    auto w_issue = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int ws = (w_k * nslab + w_slab) * w_slab_bytes;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                w[S][c][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc_w, w_lane_off + c * 2048 + j * 1024, ws, 0));
        if (++w_slab == nslab) {
            w_slab = 0;
            w_k = item_at(++w_item) & 0xff;
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------
    // byte offset of this lane's 16-byte chunk of image row li for the 16-channel group j (row block b: + 2048 b)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned foffb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) foffb[j] = lds0 + 4 * (li * KS + 4 * ((4 * j + lq) ^ ((li >> 1) & 7)));
    f32x4 a0[8], a1[8];                            // group 0 (read one stage ahead) / group 1 of the row blocks
    unsigned long long olw[8];                     // flush list words of the item (see below)
#pragma unroll
    for (int b = 0; b < 8; ++b) { a0[b] = a1[b] = f32x4{0.f, 0.f, 0.f, 0.f}; olw[b] = 0; }
    int c_slot = 0;                                // ring slot (bytes) of the stage being multiplied
    const int colb = (32 * wave + li) * 4;         // byte offset of this lane's first column inside a tile row
    const unsigned acc_base = (unsigned)(uintptr_t)(lds_ptr_t)(reinterpret_cast<char*>(acc_lds)) + colb;
    const unsigned list_base = (unsigned)(uintptr_t)(lds_ptr_t)(reinterpret_cast<char*>(out_list)) + 8 * lq;

    // Fragment reads, list reads, the flush atomics and their waits are asm: the row-block count of a stage is a
    // run-time value, and over guarded reads the compiler's wait insertion can only fall back to lgkmcnt(0) in
    // front of every read and every MFMA block (measured in the ISA), which serialises the LDS round trips; it
    // would also order any plain LDS read / atomic of the lists behind EVERY pending LDS-DMA (vmcnt).  Here the
    // two waits of a stage sit where everything they cover was requested a whole MFMA group (>= 256 cycles) ago.
#define LIDIFF_DS_READ_B128(dst, addr, off) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define LIDIFF_LGKM_WAIT(arr)                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(arr[0]), "+v"(arr[1]), "+v"(arr[2]), "+v"(arr[3]), "+v"(arr[4]), \
                 "+v"(arr[5]), "+v"(arr[6]), "+v"(arr[7]) :: "memory")

    // One stage with W set PAR for `nb` row blocks.  `last`: the item's last slab -> the look-ahead fragments belong
    // to the next item, whose row-block count is not known here (read all eight), and the item's flush list is
    // requested (list_addr).  Row blocks are the OUTER loop and every block is a guarded piece of straight-line code
    // (uniform branch), so there is ONE copy of every MFMA, every accumulator / fragment / W register has one home
    // for the whole kernel (per-count specialised bodies make the allocator shuffle dozens of registers at their
    // joins), and the code stays I-cache sized.  Inside a block the two column accumulators alternate: dependent
    // MFMAs are 64 cycles apart (latency 40).
    f32x4 acc[8][2];
    auto stage = [&](auto par_tag, int nb, bool last, unsigned list_addr) {
        constexpr int PAR = decltype(par_tag)::value;
        // stage barrier: the requests of the two younger stages (A of s+2, W of s+1: 8 instructions) stay in flight,
        // everything older has landed for this wave and, behind the barrier, for all of them; the look-ahead
        // fragments of this stage (requested during the previous stage's second MFMA group) are in registers
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]),
                       "+v"(a0[7]) :: "memory");
        a_issue();                                                             // stage s+3
        const unsigned img1 = foffb[1] + c_slot;
        const unsigned nxt0 = foffb[0] + ((c_slot + IMG) & (RING * IMG - 1));
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nb) LIDIFF_DS_READ_B128(a1[b], img1, 2048 * b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nb) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[b][e], w[PAR][c][0][e], acc[b][c], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        LIDIFF_LGKM_WAIT(a1);                      // group 1 of this stage: requested a whole MFMA group ago
        // look-ahead: group 0 of the next stage (its image was published by this stage's barrier)
        const int nbn = last ? 8 : nb;
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nbn) LIDIFF_DS_READ_B128(a0[b], nxt0, 2048 * b);
        if (last) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
                if (b < nb) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(olw[b]) : "v"(list_addr), "n"(32 * b) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (b < nb) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[b][e], w[PAR][c][1][e], acc[b][c], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
        w_issue(par_tag);                                                      // stage s+2 into the set just used
        c_slot = (c_slot + IMG) & (RING * IMG - 1);
    };

    if (n_items > 0) {
        // ---- pipeline prologue: A(0), A(1), W(0), A(2), W(1) in flight; group 0 of stage 0 in registers ----
        a_rows(item_at(0));
        w_k = item_at(0) & 0xff;
        a_issue();
        a_issue();
        w_issue(ic<0>{});
        a_issue();
        w_issue(ic<1>{});
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");      // A(0) landed everywhere
#pragma unroll
        for (int b = 0; b < 8; ++b) LIDIFF_DS_READ_B128(a0[b], foffb[0], 2048 * b);
        for (int it = 0; it < n_items; ++it) {
            const int wrd = item_at(it);
            const int k = wrd & 0xff, nb = ((wrd >> 8) + 15) >> 4;
#pragma unroll
            for (int b = 0; b < 8; ++b) { acc[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // the flush list of this offset -- out_list[k][16 b + 4 lq .. + 3]: the accumulator-tile rows (float index;
            // the dummy row behind the last pair) of this lane's four MFMA result rows of block b
            const unsigned list_addr = list_base + k * (BM * 2);
            for (int sp = 0; sp < nslab; sp += 2) {
                stage(ic<0>{}, nb, false, list_addr);
                stage(ic<1>{}, nb, sp + 2 >= nslab, list_addr);
            }
            // flush: one ds_add_f32 per element (the list words were requested a whole MFMA group ago)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(olw[0]), "+v"(olw[1]), "+v"(olw[2]), "+v"(olw[3]), "+v"(olw[4]),
                         "+v"(olw[5]), "+v"(olw[6]), "+v"(olw[7]), "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]),
                         "+v"(a0[4]), "+v"(a0[5]), "+v"(a0[6]), "+v"(a0[7]) :: "memory");
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b < nb) {
                    const unsigned lo = (unsigned)olw[b], hi = (unsigned)(olw[b] >> 32);
                    const unsigned r0 = acc_base + ((lo & 0xffff) << 2), r1 = acc_base + ((lo >> 16) << 2),
                                   r2 = acc_base + ((hi & 0xffff) << 2), r3 = acc_base + ((hi >> 16) << 2);
                    asm volatile("ds_add_f32 %0, %4\n\tds_add_f32 %1, %5\n\tds_add_f32 %2, %6\n\tds_add_f32 %3, %7\n\t"
                                 "ds_add_f32 %0, %8 offset:64\n\tds_add_f32 %1, %9 offset:64\n\t"
                                 "ds_add_f32 %2, %10 offset:64\n\tds_add_f32 %3, %11 offset:64"
                                 :: "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(acc[b][0][0]), "v"(acc[b][0][1]),
                                    "v"(acc[b][0][2]), "v"(acc[b][0][3]), "v"(acc[b][1][0]), "v"(acc[b][1][1]),
                                    "v"(acc[b][1][2]), "v"(acc[b][1][3]) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the asm-issued LDS atomics
    }
#undef LIDIFF_DS_READ_B128
#undef LIDIFF_LGKM_WAIT
    __syncthreads();                               // drains the look-ahead requests; the tile is complete

    // ---- epilogue: BN scale/shift, residual, ReLU; one coalesced float4 store per 4 channels -------------
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
        const int64_t o = (int64_t)orow[r] * p.c_out + col;
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

bool dense_kernel_applies(const ConvParams& p) {
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    return p.c_out % 128 == 0 && p.c_in_a % 32 == 0 && p.c_in_b % 32 == 0 && (p.c_in / 32) % 2 == 0 &&
           al16(p.in_a) && al16(p.in_b) && !(p.flags & LIDIFF_CONV_SPARSE_MAP) && p.k_vol <= 27;
}

int launch_fwd_dense(const ConvParams& p, hipStream_t st) {
    using namespace dense;
    auto kern = spconv_fwd_dense_kernel;
    static thread_local bool configured = false;
    if (!configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL));
        configured = true;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L_TOTAL, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

}  // namespace lidiff
