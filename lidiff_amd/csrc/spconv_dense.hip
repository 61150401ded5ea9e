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

constexpr int BM = 128, BN = 128, KS = 32;
constexpr int IMG = BM * KS * 4;                  // 16 KB per A image
constexpr int RING = 4;
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

// CB = 16-column MFMA blocks per wave: 1 -> eight waves (two per SIMD: while one wave is held up issuing its requests --
// an LDS-DMA or buffer load costs the issuing wave 40-60 cycles, more than the 32-cycle shadow of an MFMA -- its partner
// multiplies), 2 -> four waves (one per SIMD, every fragment read feeds two MFMAs).  Measured on the 256-channel layers of
// the bench scan: CB = 2 85 TFLOP/s, the tile kernel 91 (profiles/r02_dense_kernel_probe.txt).
template <int CB>
__global__ __launch_bounds__(512 / CB) __attribute__((amdgpu_waves_per_eu(2 / CB, 2 / CB)))
void spconv_fwd_dense_kernel(const ConvParams p_launch) {
    using namespace dense;
    constexpr int NW = 8 / CB, NT = 64 * NW;
    constexpr int T = 16 / NW;                    // LDS-DMA wave-instructions per wave and stage (16 x 1 KB / NW waves)
    constexpr int VM = T + 4 * CB;                // requests younger than A(s+1) at the top of stage s (see stage barrier)
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
    if (p.tail) p.tail += (int64_t)rep * p.tail_rows * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    STAMP(t_start);
#ifdef LIDIFF_CONV_PROBE
    const long long rt_start = __builtin_amdgcn_s_memrealtime();
    long long t_barrier = 0, t_flush = 0;
    const long long t_p2 = 0;
#endif

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
    const int w_lane_off = (((n0 >> 4) + CB * wave) * 512 + lane * 4) * 4;
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
    auto a_dma = [&](auto j_tag) {                 // ONE of the stage's four gather requests
        constexpr int j = decltype(j_tag)::value;
        const bool from_a = a_slab < nslab_a;
        const int cb4 = (from_a ? a_slab : a_slab - nslab_a) * (KS * 4);
        lds_ptr_t d = (lds_ptr_t)(smem + a_slot + (wave + NW * j) * 1024);
        const int voff = PROBE(1) ? (int)0x80000000 : rowoff[j];       // probe builds: no gather traffic
        if (from_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, d, 16, voff, cb4, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, d, 16, voff, cb4, 0, 0);
    };
    auto a_advance = [&]() {
        a_slot = (a_slot + IMG) & (RING * IMG - 1);
        if (++a_slab == nslab) {                   // next item
            a_slab = 0;
            a_rows(item_at(++a_item));
        } else if (a_slab == nslab_a) {            // a -> b: the row pitch changes
#pragma unroll
            for (int j = 0; j < T; ++j) rowoff[j] = rowv[j] >= 0 ? rowv[j] * (p.c_in_b * 4) + chb[j] : (int)0x80000000;
        }
    };

    // ---- W: two register sets; the halves of a set are re-loaded as soon as they are free --------------
    // stage s multiplies with set s & 1.  Its group-0 half is free after the stage's first MFMA group and takes
    // W(s+2) group 0 during the second group; its group-1 half is free when the stage ends and takes W(s+2) group 1
    // during the first MFMA group of stage s+1: requests are single instructions between MFMAs, never a burst.
    f32x4 w[2][CB][2];                             // [set][column block][16-channel group]
    struct WCur { int item, slab, k; };
    WCur w1{0, 0, 0}, w2{0, 0, 0};                 // group-1 half of stage s+1 / group-0 half of stage s+2
    auto w_load = [&](auto set_tag, auto c_tag, auto j_tag, const WCur& cur) {
        constexpr int S = decltype(set_tag)::value, C = decltype(c_tag)::value, J = decltype(j_tag)::value;
        const int ws = PROBE(2) ? 0 : (cur.k * nslab + cur.slab) * w_slab_bytes;      // probe: one hot slab
        w[S][C][J] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rsrc_w, w_lane_off + C * 2048 + J * 1024, ws, 0));
    };
    auto w_adv = [&](WCur& cur) {
        if (++cur.slab == nslab) {
            cur.slab = 0;
            cur.k = item_at(++cur.item) & 0xff;
        }
    };

    // ---- fragments ---------------------------------------------------------------------------------
    // byte offset of this lane's 16-byte chunk of image row li for the 16-channel group j (row block b: + 2048 b)
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
    unsigned foffb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) foffb[j] = lds0 + 4 * (li * KS + 4 * ((4 * j + lq) ^ ((li >> 1) & 7)));
    f32x4 a0[8], a1[8];                            // 16-channel groups 0 / 1 of the row blocks, read one STAGE ahead
    unsigned long long olw[8];                     // flush list words of the item (see below)
#pragma unroll
    for (int b = 0; b < 8; ++b) { a0[b] = a1[b] = f32x4{0.f, 0.f, 0.f, 0.f}; olw[b] = 0; }
    int c_slot = 0;                                // ring slot (bytes) of the stage being multiplied
    const int colb = (16 * CB * wave + li) * 4;    // byte offset of this lane's first column inside a tile row
    const unsigned acc_base = (unsigned)(uintptr_t)(lds_ptr_t)(reinterpret_cast<char*>(acc_lds)) + colb;
    const unsigned list_base = (unsigned)(uintptr_t)(lds_ptr_t)(reinterpret_cast<char*>(out_list)) + 8 * lq;

    // LIDIFF_BLOCKS(first, n, F): F(b) for b = first .. n-1 as NESTED ifs: the executed blocks are one fall-through
    // line with a single forward exit branch (a guard around every block costs two taken branches per block when the
    // compiler moves it out of line; a fall-through switch is turned into predicate masks by the CFG structuriser)
#define LIDIFF_BLOCKS_FROM1(n, F)                                                                      \
     if (__builtin_expect((n) > 1, 1)) { F(1);                                                         \
      if (__builtin_expect((n) > 2, 1)) { F(2);                                                        \
       if (__builtin_expect((n) > 3, 1)) { F(3);                                                       \
        if ((n) > 4) { F(4);                                                                           \
         if ((n) > 5) { F(5);                                                                          \
          if ((n) > 6) { F(6);                                                                         \
           if ((n) > 7) { F(7); } } } } } } }
#define LIDIFF_BLOCKS(n, F) if (__builtin_expect((n) > 0, 1)) { F(0); LIDIFF_BLOCKS_FROM1(n, F) }
    // Fragment reads, list reads, the flush and their waits are asm (the waits carry NO register operands -- tied
    // operands make the allocator copy the still in-flight registers in front of the wait -- and are pinned by
    // sched_barrier; tools/check_asm_regs.py scans the generated ISA for any read of an in-flight register): the row-block count of a stage is a run-time
    // value, and over guarded reads the compiler's wait insertion can only fall back to lgkmcnt(0) in front of
    // every read and every MFMA block (seen in the ISA), which serialises the LDS round trips; it would also order
    // a plain LDS read of the lists behind EVERY pending LDS-DMA (vmcnt).  LDS operations of a wave retire in order,
    // so the two counted waits of a stage (lgkmcnt(1): everything but the newest request) cover exactly the
    // fragments the next MFMA group needs, all requested at least one MFMA block (>= 256 cycles) earlier.
#define LIDIFF_DS_READ_B128(dst, addr, off) \
    if (!PROBE(32)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define SB __builtin_amdgcn_sched_barrier(0)
#define MM(P_, b, j, e, c)                                                                                        \
    acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x4f32((j ? a1 : a0)[b][e], w[P_][c][j][e], acc[b][c], 0, 0, 0)

    // One stage with W set P for `nb` row blocks.  Row blocks are the OUTER loop; inside a block the two column
    // accumulators alternate (dependent MFMAs 64 cycles apart, latency 40).  Block 0 -- every stage has it -- carries
    // the stage's requests, ONE between two MFMAs each: the four gather DMAs of stage s+3 and W(s+1) group 1 in the
    // first MFMA group, W(s+2) group 0 in the second.  After the MFMAs of block b the fragments of block b of the
    // NEXT stage are requested into the registers just consumed (its image was published by this stage's barrier).
    // `last`: the item's last slab -> the next stage belongs to the next item, whose row-block count is not known
    // here: the remaining blocks' fragments are requested too, and the item's flush list (list_addr).
    f32x4 acc[8][CB];
    auto stage = [&](auto par_tag, int nb, bool last, unsigned list_addr) {
        constexpr int P = decltype(par_tag)::value, Q = P ^ 1;
        STAMP(tb0);
        // stage barrier: A(s+1) has landed for this wave (VM younger requests stay in flight: the rest of stage s-2
        // and all of stage s-1) and, behind the barrier, for all of them; group 0 of this stage's fragments is in
        // registers (only the newest LDS request, a group-1 fragment, may still be on its way)
        if (PROBE(64)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(1)" :: "n"(VM) : "memory");      // probe: no barrier
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(1)\n\ts_barrier" :: "n"(VM) : "memory");
#ifdef LIDIFF_CONV_PROBE
        if (wave >= NW / 2)                        // probe: phase shift of the second wave of every SIMD (bits 8..15 x 16 cycles)
            for (int d = (p.probe >> 8) & 0xff; d > 0; --d) asm volatile("s_nop 15");
        if (((p.probe >> 16) & 3) == 2) {          // probe: the two waves of a SIMD take turns in having issue priority
            if ((wave < NW / 2) == (P == 0)) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        SB;
#ifdef LIDIFF_CONV_PROBE
        t_barrier += __builtin_readcyclecounter() - tb0;
#endif
        const unsigned nslot = (c_slot + IMG) & (RING * IMG - 1);
        const unsigned nxt0 = foffb[0] + nslot, nxt1 = foffb[1] + nslot;
#define RD0(b) LIDIFF_DS_READ_B128(a0[b], nxt0, 2048 * (b))
#define RD1(b) LIDIFF_DS_READ_B128(a1[b], nxt1, 2048 * (b))
#define BLK0(b) { _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int c = 0; c < CB; ++c) MM(P, b, 0, e, c); RD0(b); }
#define BLK1(b) { _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int c = 0; c < CB; ++c) MM(P, b, 1, e, c); RD1(b); }
        // ---- first MFMA group
        if constexpr (CB == 2) {
            MM(P, 0, 0, 0, 0); a_dma(ic<0>{}); SB;
            MM(P, 0, 0, 0, 1); a_dma(ic<1>{}); SB;
            MM(P, 0, 0, 1, 0); a_dma(ic<2>{}); SB;
            MM(P, 0, 0, 1, 1); a_dma(ic<3>{}); SB;
            MM(P, 0, 0, 2, 0); w_load(ic<Q>{}, ic<0>{}, ic<1>{}, w1); SB;
            MM(P, 0, 0, 2, CB - 1); w_load(ic<Q>{}, ic<CB - 1>{}, ic<1>{}, w1); SB;
            MM(P, 0, 0, 3, 0); MM(P, 0, 0, 3, CB - 1);
        } else {
            MM(P, 0, 0, 0, 0); a_dma(ic<0>{}); SB;
            MM(P, 0, 0, 1, 0); a_dma(ic<T - 1>{}); SB;
            MM(P, 0, 0, 2, 0); w_load(ic<Q>{}, ic<0>{}, ic<1>{}, w1); SB;
            MM(P, 0, 0, 3, 0);
        }
        RD0(0);
        SB;
        a_advance();
        w_adv(w1);
        SB;
        LIDIFF_BLOCKS_FROM1(nb, BLK0)
        if (last) {
#define TAIL0(b) if (nb <= b) RD0(b);
            TAIL0(1) TAIL0(2) TAIL0(3) TAIL0(4) TAIL0(5) TAIL0(6) TAIL0(7)
#undef TAIL0
#define RDL(b) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(olw[b]) : "v"(list_addr), "n"(32 * (b)) : "memory");
            LIDIFF_BLOCKS(nb, RDL)
#undef RDL
        }
        SB;
        // group 1 of this stage was requested during the previous stage's second MFMA group
        asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        SB;
        // ---- second MFMA group
        if constexpr (CB == 2) {
            MM(P, 0, 1, 0, 0); MM(P, 0, 1, 0, CB - 1); w_load(ic<P>{}, ic<0>{}, ic<0>{}, w2); SB;
            MM(P, 0, 1, 1, 0); MM(P, 0, 1, 1, CB - 1); w_load(ic<P>{}, ic<CB - 1>{}, ic<0>{}, w2); SB;
            MM(P, 0, 1, 2, 0); MM(P, 0, 1, 2, CB - 1); MM(P, 0, 1, 3, 0); MM(P, 0, 1, 3, CB - 1);
        } else {
            MM(P, 0, 1, 0, 0); w_load(ic<P>{}, ic<0>{}, ic<0>{}, w2); SB;
            MM(P, 0, 1, 1, 0); MM(P, 0, 1, 2, 0); MM(P, 0, 1, 3, 0);
        }
        RD1(0);
        SB;
        w_adv(w2);
        SB;
        LIDIFF_BLOCKS_FROM1(nb, BLK1)
        if (last) {
#define TAIL1(b) if (nb <= b) RD1(b);
            TAIL1(1) TAIL1(2) TAIL1(3) TAIL1(4) TAIL1(5) TAIL1(6) TAIL1(7)
#undef TAIL1
        }
        SB;
#undef RD0
#undef RD1
#undef BLK0
#undef BLK1
        c_slot = nslot;
    };

    STAMP(t_loop);
#ifdef LIDIFF_CONV_PROBE
    if (((p.probe >> 16) & 3) == 1 && wave < NW / 2) __builtin_amdgcn_s_setprio(3);   // probe: first wave of a SIMD first
    if (((p.probe >> 16) & 3) == 3 && wave >= NW / 2) __builtin_amdgcn_s_setprio(3);
#endif
    if (n_items > 0) {
        // ---- pipeline prologue: the request stream of two virtual stages in front of stage 0 ---------------
        //   A(0) | "stage -2": A(1), two dummies (W(0) group 1, loaded again below), W(0) group 0
        //        | "stage -1": A(2), W(0) group 1, W(1) group 0
        // so that every counted wait of the loop holds from stage 0 on.
        a_rows(item_at(0));
        w1.k = w2.k = item_at(0) & 0xff;
        auto a_all = [&]() {
            a_dma(ic<0>{}); a_dma(ic<T - 1>{});
            if constexpr (T == 4) { a_dma(ic<1>{}); a_dma(ic<2>{}); }
            a_advance();
        };
        auto w_half = [&](auto set_tag, auto j_tag, WCur& cur, bool adv) {
            w_load(set_tag, ic<0>{}, j_tag, cur);
            if constexpr (CB == 2) w_load(set_tag, ic<CB - 1>{}, j_tag, cur);
            if (adv) w_adv(cur);
        };
        a_all();                                                                              // A(0)
        a_all();                                                                              // A(1)
        w_half(ic<0>{}, ic<1>{}, w1, false);                                                  // (dummies)
        w_half(ic<0>{}, ic<0>{}, w2, true);                                                   // W(0) group 0
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(2 * CB) : "memory");         // A(0), A(1) landed everywhere
#pragma unroll
        for (int b = 0; b < 8; ++b) { LIDIFF_DS_READ_B128(a0[b], foffb[0], 2048 * b); }
#pragma unroll
        for (int b = 0; b < 8; ++b) { LIDIFF_DS_READ_B128(a1[b], foffb[1], 2048 * b); }
        a_all();                                                                              // A(2)
        w_half(ic<0>{}, ic<1>{}, w1, true);                                                   // W(0) group 1
        w_half(ic<1>{}, ic<0>{}, w2, true);                                                   // W(1) group 0
        for (int it = 0; it < n_items; ++it) {
            const int wrd = item_at(it);
            const int k = wrd & 0xff, nb = ((wrd >> 8) + 15) >> 4;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the flush list of this offset -- out_list[k][16 b + 4 lq .. + 3]: the accumulator-tile rows (float index;
            // the dummy row behind the last pair) of this lane's four MFMA result rows of block b
            const unsigned list_addr = list_base + k * (BM * 2);
            for (int sp = 0; sp < nslab; sp += 2) {
                stage(ic<0>{}, nb, false, list_addr);
                stage(ic<1>{}, nb, sp + 2 >= nslab, list_addr);
            }
            // ---- flush: tile[row of pair][col] += acc, read - add - write through the list (each output row occurs
            // at most once per offset and a wave owns its columns: race free; offsets in ascending order: deterministic).
            // The list words were requested before the last MFMA group; the 8 group-1 look-ahead fragments after them.
            STAMP(tf0);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            SB;
            const int nbf = PROBE(16) ? 0 : nb;
            // two blocks at a time (24 temporaries): the fragment / list registers written by the asm reads above must
            // never be spilled or copied by the compiler before their wait -- it believes they are valid at once
            unsigned ra[2][4];
            float t[2][4 * CB];
#define FL_RD(b)                                                                                                  \
    {                                                                                                             \
        const unsigned lo = (unsigned)olw[H + b], hi = (unsigned)(olw[H + b] >> 32);                              \
        ra[b][0] = acc_base + ((lo & 0xffff) << 2); ra[b][1] = acc_base + ((lo >> 16) << 2);                      \
        ra[b][2] = acc_base + ((hi & 0xffff) << 2); ra[b][3] = acc_base + ((hi >> 16) << 2);                      \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
            asm volatile("ds_read_b32 %0, %1" : "=v"(t[b][r]) : "v"(ra[b][r]) : "memory");                       \
            if constexpr (CB == 2)                                                                                \
                asm volatile("ds_read_b32 %0, %1 offset:64" : "=v"(t[b][4 * (CB - 1) + r]) : "v"(ra[b][r]) : "memory"); \
        }                                                                                                         \
    }
#define FL_WR(b)                                                                                                  \
    {                                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                           \
            const float v0 = t[b][r] + acc[H + b][0][r];                                                          \
            asm volatile("ds_write_b32 %0, %1" :: "v"(ra[b][r]), "v"(v0) : "memory");                            \
            if constexpr (CB == 2) {                                                                              \
                const float v1 = t[b][4 * (CB - 1) + r] + acc[H + b][CB - 1][r];                                  \
                asm volatile("ds_write_b32 %0, %1 offset:64" :: "v"(ra[b][r]), "v"(v1) : "memory");              \
            }                                                                                                     \
        }                                                                                                         \
    }
#define FL_WAIT(b)
#define FL_PAIR(h)                                                                                                \
    if (nbf > (h)) {                                                                                              \
        constexpr int H = (h);                                                                                    \
        FL_RD(0) if (nbf > (h) + 1) FL_RD(1)                                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        FL_WR(0) if (nbf > (h) + 1) FL_WR(1)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    }
            FL_PAIR(0) FL_PAIR(2) FL_PAIR(4) FL_PAIR(6)
#undef FL_PAIR
#undef FL_RD
#undef FL_WR
#undef FL_WAIT
#ifdef LIDIFF_CONV_PROBE
            t_flush += __builtin_readcyclecounter() - tf0;
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the asm-issued LDS writes
    }
    STAMP(t_epi);
#undef LIDIFF_DS_READ_B128
#undef LIDIFF_BLOCKS
#undef LIDIFF_BLOCKS_FROM1
#undef SB
#undef MM
    __syncthreads();                               // drains the look-ahead requests; the tile is complete

    // ---- epilogue: BN scale/shift, residual, ReLU; one coalesced float4 store per 4 channels -------------
    for (int e = tid; e < rows_here * (BN / 4); e += NT) {
        const int r = e / (BN / 4), cq = e % (BN / 4);
        const int col = n0 + 4 * cq;
        float4 v = reinterpret_cast<const float4*>(acc_lds)[r * (BN / 4) + cq];
        if (p.tail) {                              // contributions computed elsewhere (the non-centre offsets), fixed order
            const int orw = orow[r];
            for (int q = p.tail_ptr[orw], qe = p.tail_ptr[orw + 1]; q < qe; ++q) {
                const float4 s = *reinterpret_cast<const float4*>(p.tail + (int64_t)p.tail_idx[q] * p.c_out + col);
                v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
            }
        }
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
#ifdef LIDIFF_CONV_PROBE
    if (p.timeline != nullptr && lane == 0 && wave < 2) {
        STAMP(t_end);
        long long* d = p.timeline + ((int64_t)blockIdx.x * 2 + wave) * 10;
        d[0] = t_loop - t_start; d[1] = t_epi - t_loop; d[2] = t_end - t_epi; d[3] = t_barrier; d[4] = t_flush;
        d[5] = n_items; d[6] = nslab; d[7] = __builtin_amdgcn_s_memrealtime() - rt_start; d[8] = t_p2; d[9] = 0;
    }
#endif
}

bool dense_kernel_applies(const ConvParams& p) {
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    return p.c_out % 128 == 0 && p.c_in_a % 32 == 0 && p.c_in_b % 32 == 0 && (p.c_in / 32) % 2 == 0 &&
           al16(p.in_a) && al16(p.in_b) && !(p.flags & LIDIFF_CONV_SPARSE_MAP) && p.k_vol <= 27;
}

int launch_fwd_dense(const ConvParams& p, hipStream_t st) {
    using namespace dense;
    // flags bit LIDIFF_CONV_DENSE_ONE_WAVE: the four-wave form (one wave per SIMD, 32 columns per wave), kept for A/B runs
    const bool one_wave = (p.flags & LIDIFF_CONV_DENSE_ONE_WAVE) != 0;
    auto kern = one_wave ? spconv_fwd_dense_kernel<2> : spconv_fwd_dense_kernel<1>;
    static thread_local bool configured[2] = {false, false};
    if (!configured[one_wave]) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, L_TOTAL));
        configured[one_wave] = true;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(one_wave ? 256 : 512), L_TOTAL, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

}  // namespace lidiff
