// Sparse convolution with bf16 matrix operands and fp32 accumulation (v_mfma_f32_16x16x32_bf16), in two uses:
//   planes = 1: the mixed-precision TRAINING convolution -- BASELINE.json configs[4] "train.py diffusion training ... bf16"
//               (models.py:180-217 under autocast: GEMM operands in bf16, fp32 accumulate, fp32 master weights) -- forward
//               and, over the swapped map with W^T, the input gradient;
//   (planes = 2 / 3 of rounds 1-5 -- fp32-accurate results from split operands on THIS tile kernel -- are gone: round 6 does
//   that on wide register tiles with pre-cut operands, spconv_split3.hip; the weight packer still cuts 1..3 pieces.)
// Features are fp32 in HBM (BatchNorm, the loss, the optimizer and every consumer are fp32): the gathered rows are split /
// rounded (nearest even, v_cvt_pk_bf16_f32) on their way from the LDS image into the MFMA operand, the weights once per
// weight version when they are packed.  Round 5 (planes = 1, the training step): the rows may arrive as bf16 ALREADY -- the
// shadow copy their producer left beside the fp32 matrix (in_bf16) -- and three kernels serve them: the two-stage kernel
// below, a ring of four 32-channel stages, and (the default of kernel_size-3 / -1 layers) 256-row tiles with the accumulators in
// registers and no pair lists.
//
// Same decomposition as spconv.hip -- a workgroup owns 128 output rows x BN output channels, accumulator tile in LDS, pairs
// compacted per offset by wave ballot, gathered rows DMA'd into a source-swizzled LDS image, 16-pair row blocks -- but the
// waves form a WR x WC grid: wave (wr, wc) owns the row blocks wr, wr + WR, ... and the 32 columns [32 wc, 32 wc + 32), so
// every converted A fragment feeds two column blocks (x planes) and the A image is read WC times per stage, not 8 times:
// one bf16 MFMA is 1/15 of the fp32 MFMA time for the same work, so this kernel is bound by LDS reads, conversions and
// requests, not by the matrix pipe.
#include "spconv.h"

namespace lidiff {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// W [K, c_in, c_out] fp32 row-major -> bf16 [K][slab32][c_out/16][plane][lane 0..63][i 0..7] with
// k_in = 32 slab + 8 (lane >> 4) + i and col = 16 nt + (lane & 15) (the B operand of v_mfma_f32_16x16x32_bf16); zero beyond
// c_in.  plane 0 = bf16(w), plane 1 = bf16(w - plane 0), plane 2 = bf16(w - plane 0 - plane 1) (round to nearest even).
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, int k_vol, int c_in, int c_out, int nslab, int planes,
                                         __bf16* __restrict__ wp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    int64_t rest = idx >> 9;
    const int plane = (int)(rest % planes);
    rest /= planes;
    const int nt16 = c_out >> 4;
    const int nt = (int)(rest % nt16);
    rest /= nt16;
    const int slab = (int)(rest % nslab);
    const int k = (int)(rest / nslab);
    const int kin = 32 * slab + 8 * (lane >> 4) + i;
    const int col = 16 * nt + (lane & 15);
    float v = kin < c_in ? w[((int64_t)k * c_in + kin) * c_out + col] : 0.f;
    __bf16 piece = (__bf16)v;
    for (int q = 0; q < plane; ++q) {
        v -= (float)piece;
        piece = (__bf16)v;
    }
    wp[idx] = piece;
}

// A workgroup owns BM (128 or 64) output rows x 32 WC columns; WC x WR waves: wave (wr, wc) owns the 16-row blocks
// wr + WR j (j < RB = BM / 16 / WR) and the columns [32 wc, 32 wc + 32).  (BM = 64 -- 79 KB of LDS, two workgroups per CU --
// was measured and takes the SAME time as BM = 128 on every layer (profiles/r02_bf16_kernel_probe.txt): the kernel is bound
// by per-CU throughput, not by the phases of one workgroup; only BM = 128 is instantiated.)
// ABF: the feature rows are bf16 ALREADY (the shadow copy their producer -- or lidiff_cast_bf16 -- left beside the fp32 matrix):
// half the gather requests and bytes, an image row of KS channels is KS * 2 bytes, a lane's MFMA operand (8 channels of one pair
// row) is ONE 16-byte read and needs no conversion.  Same pairs, same products, same order of sums: bit-identical to rounding the
// fp32 rows on the fly (planes = 1 only).
template <int BM, int WC, int WR, int KS, int P, bool ABF = false>
__global__ __launch_bounds__(64 * WC * WR, BM == 64 ? 2 : 1) void spconv_fwd_bf16_kernel(const ConvParams p_launch) {
    static_assert(!ABF || P == 1, "bf16 rows: one plane");
    constexpr int BN = 32 * WC, NW = WC * WR, NT = 64 * NW, RB = BM / 16 / WR;
    constexpr int EB = ABF ? 2 : 4;              // bytes per stored feature element
    constexpr int AF = BM * KS * EB / 4;         // floats per A image
    constexpr int NCHK = KS * EB / 16, RPI = 64 / NCHK, NINST = BM / RPI, T = (NINST + NW - 1) / NW;
    static_assert(NCHK == 4 || NCHK == 8 || NCHK == 16, "image rows of 64, 128 or 256 bytes");
    // 16-byte chunks of an image row are XOR-permuted by the row so that the 16 rows a quarter wave reads at one channel offset
    // fall into 16 different bank groups (256 bytes of banks = 1, 2 or 4 image rows)
    // (64-byte rows: ds_read_b128 serves lanes {0-3, 12-15, 20-27} -- not 16 consecutive ones -- in one LDS cycle, i.e. rows
    // {0-3, 12-15} at channel group lq and rows {4-11} at lq + 1: XOR by (r >> 2) & 2 keeps them apart, (r >> 2) & 3 does not
    // -- SQ_LDS_BANK_CONFLICT 37 % of the LDS cycles of the wide kernel with the latter)
    auto swz = [](int r) { return NCHK == 16 ? r & 15 : NCHK == 8 ? (r >> 1) & 7 : (r >> 2) & 2; };
    constexpr int NS = KS / 32;                  // 32-channel steps per stage
    // ABF: the stage's W fragments ([NS steps][BN / 16 column blocks] of 1 KB) come through LDS by DMA as well, ONE copy per
    // workgroup -- each of the WR row groups loaded its own copy into registers before: 16 instead of 32 vector-memory requests
    // per 64-channel stage beside the 16 of the bf16 rows (the request stream is what bounds this kernel)
    // the accumulator tile's 16-byte chunks are XOR-permuted by the row's low bits (see the flush): the largest power of two
    // dividing the chunks per row, at most 16
    constexpr int TCH = BN / 4;
    constexpr int TSWZ = (TCH % 16 == 0) ? 16 : (TCH % 8 == 0) ? 8 : (TCH % 4 == 0) ? 4 : 2;
    constexpr bool WLDS = ABF;
    constexpr int WBLK = NS * (BN / 16);         // 1 KB blocks per W image
    constexpr int WIMG = WLDS ? WBLK * 1024 : 0; // bytes per W image
    ConvParams p = p_launch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* a_buf = reinterpret_cast<float*>(smem);                       // 2 images
    char* w_img = reinterpret_cast<char*>(a_buf + 2 * AF);               // 2 W images (WLDS)
    float* acc_lds = a_buf + 2 * AF + 2 * WIMG / 4;                      // (BM + 1) x BN
    int32_t* in_list = reinterpret_cast<int32_t*>(acc_lds + (BM + 1) * BN);
    int32_t* out_list = in_list + p.k_vol * BM;                          // float index of the accumulator row
    int32_t* cnt = out_list + p.k_vol * BM;
    int32_t* orow = cnt + 32;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tiles_all = p.tiles_m * p.replicas;
    const int tmr = (g / p.tiles_n) * 8 + xcd;
    if (tmr >= tiles_all) return;
    const int rep = tmr / p.tiles_m, tm = tmr - rep * p.tiles_m;
    p.in_a = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_a) + (int64_t)rep * p.m_in * p.c_in_a * EB);
    if (p.in_b) p.in_b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_b) + (int64_t)rep * p.m_in * p.c_in_b * EB);
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wr = wave / WC;
    const int li = lane & 15, lq = lane >> 4;

    // ---- pair lists ----------------------------------------------------------------------------------------
    for (int e = tid; e < (BM + 1) * BN / 4; e += NT) reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = tid; r < rows_here; r += NT) orow[r] = (int32_t)(row0 + r);
    if (p.nbr == nullptr) {
        for (int r = tid; r < BM; r += NT) {
            in_list[r] = (int32_t)min(row0 + r, p.m_out - 1);
            out_list[r] = r < rows_here ? r * BN : BM * BN;
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        int32_t* raw = reinterpret_cast<int32_t*>(a_buf);
        static_assert(27 * BM * 4 <= 2 * AF * 4, "raw neighbour block must fit in the A images");
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
                    out_list[k * BM + q] = r * BN;
                }
                pos += __popcll(m);
            }
#pragma unroll
            for (int c = 0; c < BM; c += 64)
                if (c + lane >= pos) out_list[k * BM + c + lane] = BM * BN;      // dummy row
            if (lane == 0) cnt[k] = pos;
        }
    }
    __syncthreads();

    const int nslab = (p.c_in + KS - 1) / KS;
    const int nslab32 = (p.c_in + 31) / 32;
    const int nt16 = p.c_out >> 4;
    const int w_slab_bytes = nt16 * P * 1024;                           // one 32-channel slab of one offset: nt16 x P planes x 1 KB
    const int w_lane_off = (((n0 >> 4) + 2 * wc) * P * 64 + lane) * 16;
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.wp), 0, (int)((size_t)p.k_vol * nslab32 * w_slab_bytes), 0x00020000);
    int chb[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const int r = RPI * (wave + NW * j) + lane / NCHK;
        chb[j] = 16 * ((lane % NCHK) ^ swz(r));
    }
    // byte offsets of this lane's 16-byte chunks (channels 32 s + 8 lq .. + 7 of image row li: two chunks of fp32, one of bf16)
    // per 32-channel step
    int foff[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int chunk = ABF ? 4 * s + lq : 8 * s + 2 * lq + h;
            foff[s][h] = li * (KS * EB) + 16 * (chunk ^ swz(li));
        }

    struct WRegs {
        uint4 v[NS][2][P];         // [32-channel step][column block][plane]
    };
    auto issue = [&](int img, int k, int slab, int n, WRegs& w) {
        const int ws = (p.probe & 2) ? 0 : (k * nslab32 + slab * NS) * w_slab_bytes;
        if constexpr (WLDS) {
            char* wdst = w_img + (img ? WIMG : 0);
#pragma unroll
            for (int j = 0; j < (WBLK + NW - 1) / NW; ++j) {
                const int b = wave + NW * j;                           // block b = (step b / (BN / 16), column block b % (BN / 16))
                if (WBLK % NW == 0 || b < WBLK)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(wdst + b * 1024), 16,
                                                             (((n0 >> 4) + b % (BN / 16)) * 64 + lane) * 16,
                                                             ws + (b / (BN / 16)) * w_slab_bytes, 0, 0);
            }
        } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < P; ++q)
                    w.v[s][cb][q] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsrc_w, w_lane_off + (cb * P + q) * 1024, ws + s * w_slab_bytes, 0));
        }
        const int k0 = slab * KS;
        const bool from_a = k0 < p.c_in_a;
        const float* src = from_a ? p.in_a : p.in_b;
        const int cw4 = (from_a ? p.c_in_a : p.c_in_b) * EB;                               // row pitch in bytes
        const int cb4 = (from_a ? k0 : k0 - p.c_in_a) * EB;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(p.m_in * cw4), 0x00020000);
        char* dst = reinterpret_cast<char*>(a_buf) + img;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int t = wave + NW * j;
            if (T * NW == NINST || t < NINST) {
                const int r = RPI * t + lane / NCHK;
                const int row = r < n && !(p.probe & 1) ? in_list[k * BM + r] : -1;
                const int voff = row >= 0 ? row * cw4 + chb[j] : (int)0x80000000;          // OOB -> zero fill
                if (p.probe & 16) continue;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + t * 1024), 16, voff, cb4, 0, 0);
            }
        }
    };

    // ---- main loop: (offset, slab) stages, requests one stage ahead, one barrier per stage ----------------------
    // The compiler does not make s_barrier wait for LDS-DMA and puts `s_waitcnt vmcnt(0)` in front of any LDS read it can
    // see while a DMA is in flight (which would serialise the prefetch), so the stage barrier and the fragment reads are
    // inline asm with their own waits; tools/check_asm_regs.py checks on the listing that nothing touches a fragment
    // register between its ds_read and the wait.
#define LIDIFF_STAGE_BARRIER()                                                          \
    do {                                                                                \
        if (p.probe & 32) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
    } while (0)
    constexpr int IMG = AF * 4;
    int img = 0;
    WRegs wcur, wnext;
    int k_cur = 0;
    while (k_cur < p.k_vol && cnt[k_cur] == 0) ++k_cur;
    if (k_cur < p.k_vol) issue(0, k_cur, 0, cnt[k_cur], wcur);
    LIDIFF_STAGE_BARRIER();
    const lds_ptr_t a_lds = (lds_ptr_t)a_buf;

    // all stages of one offset for a wave with NJ active row blocks (wr, wr + WR, ...), then the flush
    auto run_offset = [&](auto nj_tag, int n, int k_next) {
        constexpr int NJ = decltype(nj_tag)::value;
        constexpr int NI = NJ * NS;                                   // (row block, 32-channel step) items per stage
        f32x4 acc[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int slab = 0; slab < nslab; ++slab) {
            if constexpr (WLDS) {                                     // this stage's fragments: landed before the barrier behind us
                const char* wsrc = w_img + (img ? WIMG : 0) + lane * 16;
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        wcur.v[s][cb][0] = *reinterpret_cast<const uint4*>(wsrc + (s * (BN / 16) + 2 * wc + cb) * 1024);
            }
            if (slab + 1 < nslab) issue(img ^ IMG, k_cur, slab + 1, n, wnext);
            else if (k_next < p.k_vol) issue(img ^ IMG, k_next, 0, cnt[k_next], wnext);
            if (NJ > 0 && !(p.probe & 4)) {
                const unsigned base = (unsigned)(uintptr_t)a_lds + img;
                f32x4 f[2][2];                                        // [item parity][first / second 16-byte chunk]
                auto read = [&](int it) {
                    const int jj = it / NS, ss = it % NS;
                    const unsigned a0 = base + (wr + WR * jj) * (16 * KS * EB);
                    const unsigned x0 = a0 + foff[ss][0], x1 = a0 + foff[ss][1];
                    f32x4 r0, r1;
                    if constexpr (ABF) {
                        asm volatile("ds_read_b128 %0, %1" : "=&v"(r0) : "v"(x0));
                        r1 = r0;
                    } else {
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1));
                    }
                    f[it & 1][0] = r0;
                    f[it & 1][1] = r1;
                };
                read(0);
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int jj = it / NS, ss = it % NS;
                    if (it + 1 < NI) {
                        read(it + 1);
                        if constexpr (ABF) asm volatile("s_waitcnt lgkmcnt(1)");
                        else asm volatile("s_waitcnt lgkmcnt(2)");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 lo = f[it & 1][0], hi = f[it & 1][1];
                    f32x8 rest = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf16x8 a[P];
                    if constexpr (ABF) a[0] = __builtin_bit_cast(bf16x8, lo);      // the row's 8 channels as they are stored
#pragma unroll
                    for (int q = 0; q < (ABF ? 0 : P); ++q) {
                        a[q] = __builtin_convertvector(rest, bf16x8);              // round to nearest even
                        if (q + 1 < P) {                                          // rest -= float(a[q]): two bit ops per packed pair
                            const uint4 pk = __builtin_bit_cast(uint4, a[q]);
                            const unsigned w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                rest[2 * e] -= __builtin_bit_cast(float, w4[e] << 16);
                                rest[2 * e + 1] -= __builtin_bit_cast(float, w4[e] & 0xffff0000u);
                            }
                        }
                    }
                    // all products of pieces with i + j < P, the smallest first
#pragma unroll
                    for (int d = P - 1; d >= 0; --d)
#pragma unroll
                        for (int i = 0; i <= d; ++i)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb)
                                acc[jj][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(          // operands swapped: see the flush
                                    __builtin_bit_cast(bf16x8, wcur.v[ss][cb][d - i]), a[i], acc[jj][cb], 0, 0, 0);
                }
            }
            LIDIFF_STAGE_BARRIER();
            if constexpr (!WLDS) wcur = wnext;
            img ^= IMG;
        }
        // flush: tile[row of pair][col] += acc.  An output row occurs at most once per offset, waves of one offset own disjoint
        // (row block, column) pieces, and a stage barrier lies between the flushes of two offsets.
        // The MFMAs run with the operands SWAPPED (W fragment first): the transposed product leaves in each lane four CONSECUTIVE
        // channels (32 wc + 16 cb + 4 lq .. + 3) of ONE pair row (16 block + li) -- one 16-byte read-modify-write of the tile per
        // (block, column block) instead of four 4-byte ones to four rows; a row's 16-byte chunks are XOR-permuted by the row so
        // that the 16 rows a quarter wave flushes at one channel offset fall into different banks (as spconv.hip, round 4).
        const int32_t* ol = out_list + k_cur * BM + li;
        if (p.probe & 8) return;
        int addr[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int o = ol[16 * (wr + WR * j)];                     // float index of the pair's tile row
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) addr[j][cb] = o + 4 * ((8 * wc + 4 * cb + lq) ^ ((o / BN) & (TSWZ - 1)));
        }
        f32x4 old[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) old[j][cb] = *reinterpret_cast<const f32x4*>(acc_lds + addr[j][cb]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4*>(acc_lds + addr[j][cb]) = old[j][cb] + acc[j][cb];
    };

    while (k_cur < p.k_vol) {
        const int n = cnt[k_cur];
        const int nb = (n + 15) >> 4;
        int k_next = k_cur + 1;
        while (k_next < p.k_vol && cnt[k_next] == 0) ++k_next;
        const int nj = nb > wr ? (nb - wr + WR - 1) / WR : 0;          // this wave's active row blocks: wr + WR j < nb
        if (RB >= 4 && nj == 4) run_offset(ic<(RB >= 4 ? 4 : 0)>{}, n, k_next);
        else if (RB >= 4 && nj == 3) run_offset(ic<(RB >= 4 ? 3 : 0)>{}, n, k_next);
        else if (RB >= 2 && nj == 2) run_offset(ic<(RB >= 2 ? 2 : 0)>{}, n, k_next);
        else if (nj == 1) run_offset(ic<1>{}, n, k_next);
        else run_offset(ic<0>{}, n, k_next);
        k_cur = k_next;
    }
#undef LIDIFF_STAGE_BARRIER
    __syncthreads();                     // the last flush (no request is in flight any more)

    // ---- epilogue ------------------------------------------------------------------------------------------
    for (int e = tid; e < rows_here * (BN / 4); e += NT) {
        const int r = e / (BN / 4), cq = e % (BN / 4);
        const int col = n0 + 4 * cq;
        float4 v = reinterpret_cast<const float4*>(acc_lds)[r * (BN / 4) + (cq ^ (r & (TSWZ - 1)))];
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

template <int BM, int WC, int WR, int KS, int P, bool ABF = false>
static int launch_bf16(const ConvParams& p, hipStream_t st) {
    constexpr int BN = 32 * WC;
    const size_t lds = (size_t)2 * BM * KS * (ABF ? 2 : 4) + (ABF ? (size_t)2 * (KS / 32) * (BN / 16) * 1024 : 0) +
                       (size_t)(BM + 1) * BN * 4 + (size_t)p.k_vol * BM * 8 + 32 * 4 + BM * 4;
    auto kern = spconv_fwd_bf16_kernel<BM, WC, WR, KS, P, ABF>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WC * WR), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

template <int P>
static int dispatch_bf16(const ConvParams& p, bool ks64, hipStream_t st) {
    // three planes: 32-channel stages only (the W registers of a 64-channel stage would not fit beside the accumulators)
#define LIDIFF_BF16(BM, WC, WR) \
    return ks64 && P < 3 ? launch_bf16<BM, WC, WR, (P < 3 ? 64 : 32), P>(p, st) : launch_bf16<BM, WC, WR, 32, P>(p, st)
    if (p.c_out % 128 == 0) LIDIFF_BF16(128, 4, 2);
    if (p.c_out % 96 == 0) LIDIFF_BF16(128, 3, 2);
    if (p.c_out % 64 == 0) LIDIFF_BF16(128, 2, 4);
    LIDIFF_BF16(128, 1, 8);
#undef LIDIFF_BF16
}


// ---------------------------------------------------------------------------------------
// bf16 rows, a RING of stages (round 5).  The kernel above issues a stage's requests one stage ahead and waits for all of them at
// the stage's barrier; with bf16 rows a stage is shorter than a memory latency, so every stage waits for its own data (ablations:
// requests, flush and barrier removed, 414 of 739 us remain on 256 -> 256).  Here a stage is 32 channels (8 KB of rows + the
// [BN / 16] KB of W fragments), four of them live in LDS at once -- the same 64 KB as two 64-channel stages -- and the requests of
// stage s + 3 are issued while stage s is multiplied: the barrier of a stage waits only until the two newest stages' requests are
// the ones still in flight (s_waitcnt vmcnt(2 x requests per stage): every wave issues the same number of requests per stage, the
// surplus ones of narrow tiles repeat a block).  The LDS-DMA is issued from inline asm: the compiler does not see LDS being written
// behind its back, so it does not put a full `s_waitcnt vmcnt(0)` in front of every LDS read while requests are in flight (which
// is what it does for the builtin, and what the counted waits of the kernel above work around) -- ordering is by the stage
// barriers alone.  Same pairs, products and order of sums per accumulator as the kernel above: bit-identical results.
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16_to_lds(const i32x4 rsrc, const unsigned lds_addr, const int voff, const int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

template <int BM, int WC, int WR>
__global__ __launch_bounds__(64 * WC * WR) void spconv_fwd_bf16_ring_kernel(const ConvParams p_launch) {
    constexpr int KS = 32, R = 4, D = 3;                    // channels per stage, stages in LDS, stages the requests run ahead
    constexpr int BN = 32 * WC, NW = WC * WR, NT = 64 * NW, RB = BM / 16 / WR;
    constexpr int ABYTES = BM * KS * 2;                     // rows of a stage: 64 bytes each
    constexpr int NCHK = 4, RPI = 16, NINST = BM / RPI, T = (NINST + NW - 1) / NW;
    constexpr int WBLK = BN / 16, WBYTES = WBLK * 1024, TW = (WBLK + NW - 1) / NW;
    constexpr int PER_STAGE = T + TW;                       // requests per wave and stage
    static_assert((D - 1) * PER_STAGE <= 63, "vmcnt range");
    constexpr int TCH = BN / 4;
    constexpr int TSWZ = (TCH % 16 == 0) ? 16 : (TCH % 8 == 0) ? 8 : (TCH % 4 == 0) ? 4 : 2;
    ConvParams p = p_launch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_ring = smem;
    char* w_ring = smem + R * ABYTES;
    float* acc_lds = reinterpret_cast<float*>(smem + R * (ABYTES + WBYTES));
    int32_t* in_list = reinterpret_cast<int32_t*>(acc_lds + (BM + 1) * BN);
    int32_t* out_list = in_list + p.k_vol * BM;
    int32_t* cnt = out_list + p.k_vol * BM;
    int32_t* orow = cnt + 32;
    int32_t* act = orow + BM;                               // the offsets that have pairs, ascending; act[31] = how many

    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tmr = (g / p.tiles_n) * 8 + xcd;
    if (tmr >= p.tiles_m * p.replicas) return;
    const int rep = tmr / p.tiles_m, tm = tmr - rep * p.tiles_m;
    p.in_a = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_a) + (int64_t)rep * p.m_in * p.c_in_a * 2);
    if (p.in_b) p.in_b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_b) + (int64_t)rep * p.m_in * p.c_in_b * 2);
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave % WC, wr = wave / WC;
    const int li = lane & 15, lq = lane >> 4;

    // ---- pair lists (as the kernel above) -------------------------------------------------------------------
    for (int e = tid; e < (BM + 1) * BN / 4; e += NT) reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = tid; r < rows_here; r += NT) orow[r] = (int32_t)(row0 + r);
    if (p.nbr == nullptr) {
        for (int r = tid; r < BM; r += NT) {
            in_list[r] = (int32_t)min(row0 + r, p.m_out - 1);
            out_list[r] = r < rows_here ? r * BN : BM * BN;
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        int32_t* raw = reinterpret_cast<int32_t*>(a_ring);
        static_assert(27 * BM * 4 <= R * (ABYTES + WBYTES), "raw neighbour block must fit in the rings");
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
                    out_list[k * BM + q] = r * BN;
                }
                pos += __popcll(m);
            }
#pragma unroll
            for (int c = 0; c < BM; c += 64)
                if (c + lane >= pos) out_list[k * BM + c + lane] = BM * BN;      // dummy row
            if (lane == 0) cnt[k] = pos;
        }
    }
    __syncthreads();
    if (wave == 0) {                                        // the active offsets, in order
        const bool on = lane < p.k_vol && cnt[lane] > 0;
        const unsigned long long m = __ballot(on);
        if (on) act[popc_below(m)] = lane;
        if (lane == 0) act[31] = __popcll(m);
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(act[31]);
    const int nslab = (p.c_in + KS - 1) / KS;
    const int nst = nact * nslab;                           // stages of this tile
    const int nt16 = p.c_out >> 4;
    const int w_slab_bytes = nt16 * 1024;
    // raw buffer descriptors (base, stride 0, bytes, the flags __builtin_amdgcn_make_buffer_rsrc is given elsewhere), wave-uniform
    auto rsrc = [](const void* base, int64_t bytes) {
        const uint64_t a = (uint64_t)(uintptr_t)base;
        i32x4 d = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_readfirstlane(d[i]);
        return d;
    };
    const i32x4 rsrc_w = rsrc(p.wp, (int64_t)p.k_vol * nslab * w_slab_bytes);
    const i32x4 rsrc_a = rsrc(p.in_a, p.m_in * p.c_in_a * 2);
    const i32x4 rsrc_b = p.in_b ? rsrc(p.in_b, p.m_in * p.c_in_b * 2) : rsrc_a;
    auto swz = [](int r) { return (r >> 2) & 2; };          // four 64-byte rows per 256 bytes of banks; see the kernel above
    int tblk[T], chb[T];                                    // this lane's row block / chunk of every request (surplus ones repeat)
#pragma unroll
    for (int j = 0; j < T; ++j) {
        tblk[j] = (wave + NW * j) % NINST;
        const int r = RPI * tblk[j] + lane / NCHK;
        chb[j] = 16 * ((lane % NCHK) ^ swz(r));
    }
    const int foff = li * (KS * 2) + 16 * (lq ^ swz(li));   // this lane's 16-byte chunk (channels 8 lq .. + 7 of image row li)
    const unsigned a_base = (unsigned)(uintptr_t)(lds_ptr_t)a_ring, w_base = (unsigned)(uintptr_t)(lds_ptr_t)w_ring;

    // requests of stage `sg` = (offset act[i_oi], slab i_slab) into ring slot sg mod R; stages behind the last one request nothing
    // real (rows out of range: zero fill) so that every stage has the same number of requests
    int i_oi = 0, i_slab = 0;
    auto issue = [&](int sg) {
        const bool live = sg < nst;
        const int k = live ? act[i_oi] : 0;
        const int n = live ? cnt[k] : 0;
        const int slab = live ? i_slab : 0;
        const int slot = sg & (R - 1);
        const int ws = (k * nslab + slab) * w_slab_bytes;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int b = (wave + NW * j) % WBLK;
            dma16_to_lds(rsrc_w, w_base + slot * WBYTES + b * 1024, (((n0 >> 4) + b) * 64 + lane) * 16, ws);
        }
        const int k0 = slab * KS;
        const bool from_a = k0 < p.c_in_a;
        const int cw = (from_a ? p.c_in_a : p.c_in_b) * 2;
        const int cb = (from_a ? k0 : k0 - p.c_in_a) * 2;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int r = RPI * tblk[j] + lane / NCHK;
            const int row = r < n ? in_list[k * BM + r] : -1;
            const int voff = row >= 0 ? row * cw + chb[j] : (int)0x80000000;          // out of range -> zero fill
            dma16_to_lds(from_a ? rsrc_a : rsrc_b, a_base + slot * ABYTES + tblk[j] * 1024, voff, cb);
        }
        if (++i_slab == nslab) { i_slab = 0; ++i_oi; }
    };
    // all requests of stage sg + 1 have landed (for every wave) and are visible; the slot of stage sg is free again
#define LIDIFF_RING_BARRIER() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((D - 1) * PER_STAGE) : "memory")

    int sg = 0;                                             // the stage being multiplied
    for (int s = 0; s < D; ++s) issue(s);
    LIDIFF_RING_BARRIER();

    auto run_offset = [&](auto nj_tag, int k_cur) {
        constexpr int NJ = decltype(nj_tag)::value;
        f32x4 acc[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int slab = 0; slab < nslab; ++slab, ++sg) {
            const int slot = sg & (R - 1);
            const char* wsrc = w_ring + slot * WBYTES + lane * 16;
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wsrc + (2 * wc) * 1024);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wsrc + (2 * wc + 1) * 1024);
            issue(sg + D);
            const char* asrc = a_ring + slot * ABYTES + foff;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(asrc + (wr + WR * j) * (16 * KS * 2));
                acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, a, acc[j][0], 0, 0, 0);     // operands swapped: see the flush
                acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, a, acc[j][1], 0, 0, 0);
            }
            LIDIFF_RING_BARRIER();
        }
        // flush: one 16-byte read-modify-write of the bank-swizzled tile per (block, column block), as the kernel above
        const int32_t* ol = out_list + k_cur * BM + li;
        int addr[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int o = ol[16 * (wr + WR * j)];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) addr[j][cb] = o + 4 * ((8 * wc + 4 * cb + lq) ^ ((o / BN) & (TSWZ - 1)));
        }
        f32x4 old[NJ > 0 ? NJ : 1][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) old[j][cb] = *reinterpret_cast<const f32x4*>(acc_lds + addr[j][cb]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) *reinterpret_cast<f32x4*>(acc_lds + addr[j][cb]) = old[j][cb] + acc[j][cb];
    };

    for (int oi = 0; oi < nact; ++oi) {
        const int k_cur = act[oi];
        const int nb = (cnt[k_cur] + 15) >> 4;
        const int nj = nb > wr ? (nb - wr + WR - 1) / WR : 0;          // this wave's active row blocks: wr + WR j < nb
        if (RB >= 8 && nj > 4) {
            if (nj == 8) run_offset(ic<(RB >= 8 ? 8 : 0)>{}, k_cur);
            else if (nj == 7) run_offset(ic<(RB >= 8 ? 7 : 0)>{}, k_cur);
            else if (nj == 6) run_offset(ic<(RB >= 8 ? 6 : 0)>{}, k_cur);
            else run_offset(ic<(RB >= 8 ? 5 : 0)>{}, k_cur);
        }
        else if (RB >= 4 && nj == 4) run_offset(ic<(RB >= 4 ? 4 : 0)>{}, k_cur);
        else if (RB >= 4 && nj == 3) run_offset(ic<(RB >= 4 ? 3 : 0)>{}, k_cur);
        else if (RB >= 2 && nj == 2) run_offset(ic<(RB >= 2 ? 2 : 0)>{}, k_cur);
        else if (nj == 1) run_offset(ic<1>{}, k_cur);
        else run_offset(ic<0>{}, k_cur);
    }
#undef LIDIFF_RING_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the surplus requests of the last stages
    __syncthreads();                                       // ... and the last flush

    // ---- epilogue (as the kernel above) --------------------------------------------------------------------------
    for (int e = tid; e < rows_here * (BN / 4); e += NT) {
        const int r = e / (BN / 4), cq = e % (BN / 4);
        const int col = n0 + 4 * cq;
        float4 v = reinterpret_cast<const float4*>(acc_lds)[r * (BN / 4) + (cq ^ (r & (TSWZ - 1)))];
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

template <int BM, int WC, int WR>
static int launch_bf16_ring(const ConvParams& p, hipStream_t st) {
    constexpr int BN = 32 * WC;
    const size_t lds = (size_t)4 * (BM * 32 * 2 + (BN / 16) * 1024) + (size_t)(BM + 1) * BN * 4 + (size_t)p.k_vol * BM * 8 +
                       32 * 4 + BM * 4 + 32 * 4;
    auto kern = spconv_fwd_bf16_ring_kernel<BM, WC, WR>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WC * WR), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// bf16 rows, WIDE tiles with the accumulators in registers (round 5).  The two kernels above move 6.3 GB between the caches and
// the CUs per 256 -> 256 launch (8.4 TB/s: every 128 x 128 tile re-reads all of W and gathers its rows once per column tile) -- that
// traffic, not the matrix pipe (busy ~15 %), bounds them, and their tile cannot grow: its accumulators live in LDS because a pair
// list scatters an offset's products over the tile's rows.  With bf16 the MFMAs are cheap enough to drop the pair lists instead: a
// tile multiplies ALL of its rows for every offset -- rows without a neighbour are gathered as zeros (an out-of-range request
// moves no bytes) -- so every product lands in a fixed accumulator and the accumulators can stay in registers: a 256-row x
// BN-column tile (BN = 256 or 128) in 8 waves, wave (rg, cg) = 128 rows x BN / 4 columns = 8 x BN / 64 MFMA blocks.  Bytes per
// output element fall by half (W once per 256 rows, rows once per 256 / 128 columns); no accumulator tile, no pair lists, no flush
// in LDS -- only the ring of four 32-channel stages (rows 16 KB + W fragments BN / 16 KB each) and the tile's block of the
// neighbour table.  One fp32 sum per output over all offsets and channels (the kernels above sum per offset first): equal up to
// the order of the fp32 additions.
// (narrower layers: the 8 waves as RG row groups x 8 / RG column groups -- 96 columns: 4 x 2, a wave 64 rows x 48 columns)
template <int BN, int RG = 2>
__global__ __launch_bounds__(512) void spconv_fwd_bf16_wide_kernel(const ConvParams p_launch) {
    constexpr int BM = 256, KS = 32, R = 4, D = 3, NW = 8, NT = 512;
    constexpr int CG = NW / RG, RBW = (BM / 16) / RG;       // column groups; row blocks per wave
    constexpr int CBW = (BN / 16) / CG;                     // column blocks per wave
    static_assert(CBW * CG * 16 == BN && RBW * RG * 16 == BM, "wave grid");
    constexpr int ABYTES = BM * KS * 2;                     // 16 KB
    constexpr int NCHK = 4, RPI = 16, NINST = BM / RPI, T = NINST / NW;          // 2 row requests per wave and stage
    constexpr int WBLK = BN / 16, WBYTES = WBLK * 1024, TW = (WBLK + NW - 1) / NW;   // W requests (surplus ones repeat a block)
    constexpr int PER_STAGE = T + TW;
    ConvParams p = p_launch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_ring = smem;
    char* w_ring = smem + R * ABYTES;
    int32_t* raw = reinterpret_cast<int32_t*>(smem + R * (ABYTES + WBYTES));       // [k_vol][BM] of the neighbour table
    int32_t* act = raw + p.k_vol * BM;                                           // offsets with a neighbour in the tile; act[31] = count

    const int bid = blockIdx.x;
    const int xcd = bid & 7, g = bid >> 3;
    const int tn = g % p.tiles_n;
    const int tmr = (g / p.tiles_n) * 8 + xcd;
    if (tmr >= p.tiles_m * p.replicas) return;
    const int rep = tmr / p.tiles_m, tm = tmr - rep * p.tiles_m;
    p.in_a = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_a) + (int64_t)rep * p.m_in * p.c_in_a * 2);
    if (p.in_b) p.in_b = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.in_b) + (int64_t)rep * p.m_in * p.c_in_b * 2);
    p.out += (int64_t)rep * p.m_out * p.c_out;
    if (p.residual) p.residual += (int64_t)rep * p.m_out * p.c_out;
    const int64_t row0 = (int64_t)tm * BM;
    const int n0 = tn * BN;
    const int rows_here = (int)min((int64_t)BM, p.m_out - row0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave / CG, cg = wave % CG;
    const int li = lane & 15, lq = lane >> 4;

    // ---- the tile's block of the neighbour table; which offsets occur ---------------------------------------------
    if (tid < 32) act[tid] = 0;
    __syncthreads();
    for (int k = wave; k < p.k_vol; k += NW) {
        bool any = false;
#pragma unroll
        for (int c = 0; c < BM; c += 64) {
            const int r = c + lane;
            int v = -1;
            if (r < rows_here) v = p.nbr ? p.nbr[(int64_t)k * p.m_out + row0 + r] : (int32_t)(row0 + r);
            raw[k * BM + r] = v;
            any |= v >= 0;
        }
        if (__ballot(any) != 0ull && lane == 0) act[k] = 1;          // (flags first, compacted below)
    }
    __syncthreads();
    if (wave == 0) {
        const bool on = lane < p.k_vol && act[lane] != 0;
        const unsigned long long m = __ballot(on);          // (one wave, in lockstep: every flag is read before any is overwritten)
        if (on) act[popc_below(m)] = lane;
        if (lane == 0) act[31] = __popcll(m);
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(act[31]);
    const int nslab = (p.c_in + KS - 1) / KS;
    const int nst = nact * nslab;
    const int nt16 = p.c_out >> 4;
    const int w_slab_bytes = nt16 * 1024;
    auto rsrc = [](const void* base, int64_t bytes) {
        const uint64_t a = (uint64_t)(uintptr_t)base;
        i32x4 d = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = __builtin_amdgcn_readfirstlane(d[i]);
        return d;
    };
    const i32x4 rsrc_w = rsrc(p.wp, (int64_t)p.k_vol * nslab * w_slab_bytes);
    const i32x4 rsrc_a = rsrc(p.in_a, p.m_in * p.c_in_a * 2);
    const i32x4 rsrc_b = p.in_b ? rsrc(p.in_b, p.m_in * p.c_in_b * 2) : rsrc_a;
    auto swz = [](int r) { return (r >> 2) & 2; };          // (ds_read_b128's lane groups: see spconv_fwd_bf16_kernel)
    int chb[T];
#pragma unroll
    for (int j = 0; j < T; ++j) chb[j] = 16 * ((lane % NCHK) ^ swz(RPI * (wave + NW * j) + lane / NCHK));
    const int foff = li * (KS * 2) + 16 * (lq ^ swz(li));
    const unsigned a_base = (unsigned)(uintptr_t)(lds_ptr_t)a_ring, w_base = (unsigned)(uintptr_t)(lds_ptr_t)w_ring;

    int i_oi = 0, i_slab = 0;
    auto issue = [&](int sg) {
        const bool live = sg < nst;
        const int k = live ? act[i_oi] : 0;
        const int slab = live ? i_slab : 0;
        const int slot = sg & (R - 1);
        const int ws = (k * nslab + slab) * w_slab_bytes;
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const int b = (wave + NW * j) % WBLK;
            dma16_to_lds(rsrc_w, w_base + slot * WBYTES + b * 1024, (((n0 >> 4) + b) * 64 + lane) * 16, ws);
        }
        const int k0 = slab * KS;
        const bool from_a = k0 < p.c_in_a;
        const int cw = (from_a ? p.c_in_a : p.c_in_b) * 2;
        const int cb = (from_a ? k0 : k0 - p.c_in_a) * 2;
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int t = wave + NW * j;
            const int row = live ? raw[k * BM + RPI * t + lane / NCHK] : -1;
            const int voff = row >= 0 ? row * cw + chb[j] : (int)0x80000000;          // no neighbour: zeros, no bytes moved
            dma16_to_lds(from_a ? rsrc_a : rsrc_b, a_base + slot * ABYTES + t * 1024, voff, cb);
        }
        if (++i_slab == nslab) { i_slab = 0; ++i_oi; }
    };
#define LIDIFF_RING_BARRIER() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"((D - 1) * PER_STAGE) : "memory")

    f32x4 acc[RBW][CBW];
#pragma unroll
    for (int j = 0; j < RBW; ++j)
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < D; ++s) issue(s);
    LIDIFF_RING_BARRIER();
    for (int sg = 0; sg < nst; ++sg) {
        const int slot = sg & (R - 1);
        const char* wsrc = w_ring + slot * WBYTES + (CBW * cg) * 1024 + lane * 16;
        const char* asrc = a_ring + slot * ABYTES + (RBW * rg) * (16 * KS * 2) + foff;
        bf16x8 w[CBW], a[RBW];
#pragma unroll
        for (int c = 0; c < CBW; ++c) w[c] = *reinterpret_cast<const bf16x8*>(wsrc + c * 1024);
#pragma unroll
        for (int j = 0; j < RBW; ++j) a[j] = *reinterpret_cast<const bf16x8*>(asrc + j * (16 * KS * 2));
        issue(sg + D);
#pragma unroll
        for (int j = 0; j < RBW; ++j)
#pragma unroll
            for (int c = 0; c < CBW; ++c)
                acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[c], a[j], acc[j][c], 0, 0, 0);     // swapped: 4 channels of one row per lane
        LIDIFF_RING_BARRIER();
    }
#undef LIDIFF_RING_BARRIER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the surplus requests of the last stages

    // ---- epilogue straight from the registers: lane (li, lq) holds channels n0 + 16 (CBW cg + c) + 4 lq .. + 3 of row 16 (RBW rg + j) + li
#pragma unroll
    for (int c = 0; c < CBW; ++c) {
        const int col = n0 + 16 * (CBW * cg + c) + 4 * lq;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + col);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + col);
#pragma unroll
        for (int j = 0; j < RBW; ++j) {
            const int r = 16 * (RBW * rg + j) + li;
            if (r >= rows_here) continue;
            const int64_t o = (row0 + r) * p.c_out + col;
            float4 v = make_float4(acc[j][c][0], acc[j][c][1], acc[j][c][2], acc[j][c][3]);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
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
}

template <int BN, int RG = 2>
static int launch_bf16_wide(const ConvParams& p, hipStream_t st) {
    const size_t lds = (size_t)4 * (256 * 32 * 2 + (BN / 16) * 1024) + (size_t)p.k_vol * 256 * 4 + 32 * 4;
    auto kern = spconv_fwd_bf16_wide_kernel<BN, RG>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, 256);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div((int64_t)q.tiles_m * q.replicas, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

// bf16 feature rows (planes = 1): the same tiles and stages
static int dispatch_bf16_rows(const ConvParams& p, bool ks64, int mode, hipStream_t st) {
    // measured (tools/conv_probe.py --kernel bf16 --rows16, bench scan at sigma 1; us two-stage -> ring): 256 -> 256 at stride 8
    // 744 -> 770, at stride 16 295 -> 326, 384 -> 256 990 -> 1054, 128 -> 128 at stride 8 241 -> 253, at stride 4 307 -> 300,
    // 64 -> 64 183 -> 184, 96 -> 96 at stride 2 302 -> 247: both kernels move the same 6.3 GB per 256 -> 256 launch between L2 and
    // the CUs (8.4 TB/s: half of it W fragments, re-read by every tile), which is what bounds them once the latency is hidden --
    // the ring wins where stages are nearly empty (few pairs per offset, 32-channel slabs anyway): the 96-column tiles take it
    const bool ring = mode == 2 || (mode == 1 && p.c_out % 128 != 0 && p.c_out % 96 == 0);
    // (in_bf16: 1 = this rule, 2 = ring, 3 = two stages, 4 = wide register tiles where the width allows)
    if (mode == 4) {
        if (p.c_out % 256 == 0) return launch_bf16_wide<256>(p, st);
        if (p.c_out % 128 == 0) return launch_bf16_wide<128>(p, st);
        if (p.c_out % 96 == 0) return launch_bf16_wide<96, 4>(p, st);
        if (p.c_out % 64 == 0) return launch_bf16_wide<64, 4>(p, st);
        return launch_bf16_wide<32, 4>(p, st);
    }
    if (ring) {
        if (p.c_out % 128 == 0) return launch_bf16_ring<128, 4, 2>(p, st);
        if (p.c_out % 96 == 0) return launch_bf16_ring<128, 3, 2>(p, st);
        if (p.c_out % 64 == 0) return launch_bf16_ring<128, 2, 4>(p, st);
        return launch_bf16_ring<128, 1, 8>(p, st);
    }
#define LIDIFF_BF16(BM, WC, WR) \
    return ks64 ? launch_bf16<BM, WC, WR, 64, 1, true>(p, st) : launch_bf16<BM, WC, WR, 32, 1, true>(p, st)
    if (p.c_out % 128 == 0) LIDIFF_BF16(128, 4, 2);
    if (p.c_out % 96 == 0) LIDIFF_BF16(128, 3, 2);
    if (p.c_out % 64 == 0) LIDIFF_BF16(128, 2, 4);
    LIDIFF_BF16(128, 1, 8);
#undef LIDIFF_BF16
}

// =======================================================================================
// Weight gradient with bf16 operands (bf16 training): dW[k] = in[pairs_k]^T g[pairs_k] as in spconv.hip's
// spconv_bwd_w_kernel -- same tiles ([16 NBI ci] x [128 CB co] of dW[k] in MFMA accumulators, wave w the co blocks
// w CB .., pair slices, workspace reduction) -- but the gathered rows of a 64-pair chunk are rounded to bf16 and stored
// TRANSPOSED in LDS, a_t[ci][pair] / g_t[co][pair] (row pitch 144 B: conflict-free 16-byte reads), so the operand of
// v_mfma_f32_16x16x32_bf16 -- 8 consecutive pairs of one channel -- is one ds_read_b128 and a chunk is 2 MFMA steps per
// block instead of 16.  Each thread gathers one (channel quad, pair octet): 8 float4 loads, 4 ds_write_b128.
// ABF: `in` and `g` are bf16 rows already (8-byte loads of 4 channels, no rounding here): the same operands, bit-identical sums.
template <int NBI, int CB, bool IDENT, bool ABF = false>
__global__ __launch_bounds__(512) void spconv_bwd_w_bf16_kernel(const float* __restrict__ in_a, int c_in_a,
                                                                const float* __restrict__ in_b, int c_in_b,
                                                                const float* __restrict__ g, const int32_t* __restrict__ pairs_in,
                                                                const int32_t* __restrict__ pairs_out,
                                                                const int32_t* __restrict__ offset_ptr, int64_t m_out,
                                                                int c_out, int64_t per, float* __restrict__ dw,
                                                                float* __restrict__ part, int k_vol) {
    constexpr int CIT = 16 * NBI, COT = 128 * CB;
    constexpr int PB = 2 * kDwPairs + 16;                    // LDS row pitch in bytes
    constexpr int UA = (CIT / 4) * 8, UG = (COT / 4) * 8;    // (channel quad, pair octet) units per chunk
    constexpr int NA = (UA + 511) / 512, NG = (UG + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* a_t = smem;                                        // [ci][pair] bf16
    char* g_t = smem + CIT * PB;                             // [co][pair] bf16
    const int c_in = c_in_a + c_in_b;
    const int co_tiles = (c_out + COT - 1) / COT;
    const int ci0 = (blockIdx.y / co_tiles) * CIT, co0 = (blockIdx.y % co_tiles) * COT;
    const int slot = (int)(blockIdx.z * gridDim.x + blockIdx.x);       // slots by pair count: see spconv.h
    int k, local;
    int64_t p_lo, p_hi;
    if (!dw_slot_offset(IDENT ? nullptr : offset_ptr, k_vol, m_out, per, slot, k, local, p_lo, p_hi)) return;       // idle slot
    const int64_t s_lo = p_lo + (int64_t)local * per, s_hi = min(p_hi, s_lo + per);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;

    using Quad = std::conditional_t<ABF, uint2, float4>;     // 4 channels of one row as stored
    auto load4 = [](const float* base, unsigned elem) {
        if constexpr (ABF) return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(base) + (size_t)elem * 2);
        else return *reinterpret_cast<const float4*>(base + elem);
    };
    Quad pa[NA][8], pg[NG][8];                               // the next chunk's rows, in flight
    unsigned keep_a[NA], keep_g[NG];                         // bit i: pair i of the unit is real
    auto fetch = [&](int64_t base) {
        int ra[NA][8], rg[NG][8];
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int oct = min(tid + 512 * t, UA - 1) / (CIT / 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t pp = min(base + 8 * oct + i, s_hi - 1);
                ra[t][i] = IDENT ? (int)pp : pairs_in[pp];
            }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int oct = min(tid + 512 * t, UG - 1) / (COT / 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t pp = min(base + 8 * oct + i, s_hi - 1);
                rg[t][i] = IDENT ? (int)pp : pairs_out[pp];
            }
        }
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int u = tid + 512 * t, uc = min(u, UA - 1), oct = uc / (CIT / 4), ci = ci0 + (uc % (CIT / 4)) * 4;
            const int cic = min(ci, c_in - 4);
            keep_a[t] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pa[t][i] = cic < c_in_a ? load4(in_a, (unsigned)(ra[t][i] * c_in_a + cic))
                                        : load4(in_b, (unsigned)(ra[t][i] * c_in_b + (cic - c_in_a)));
                keep_a[t] |= (u < UA && base + 8 * oct + i < s_hi && ci < c_in) ? 1u << i : 0u;
            }
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int u = tid + 512 * t, uc = min(u, UG - 1), oct = uc / (COT / 4), co = co0 + (uc % (COT / 4)) * 4;
            keep_g[t] = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pg[t][i] = load4(g, (unsigned)(rg[t][i] * c_out + min(co, c_out - 4)));
                keep_g[t] |= (u < UG && base + 8 * oct + i < s_hi && co < c_out) ? 1u << i : 0u;
            }
        }
    };
    // the unit's 8 pairs x 4 channels -> four rows of 8 bf16 (round to nearest even), zero where the pair is not real
    auto put = [&](char* dst, const Quad (&v)[8], unsigned keep, int quad, int oct) {
        if constexpr (ABF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned h[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned word = j < 2 ? v[i].x : v[i].y;
                    h[i] = (keep >> i) & 1u ? ((j & 1) ? word >> 16 : word & 0xffffu) : 0u;
                }
                *reinterpret_cast<uint4*>(dst + (4 * quad + j) * PB + 16 * oct) =
                    make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            }
        } else {
            float m[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) m[i] = (keep >> i) & 1u ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x8 r;
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = (j == 0 ? v[i].x : j == 1 ? v[i].y : j == 2 ? v[i].z : v[i].w) * m[i];
                *reinterpret_cast<bf16x8*>(dst + (4 * quad + j) * PB + 16 * oct) = __builtin_convertvector(r, bf16x8);
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int u = tid + 512 * t;
            if (u < UA) put(a_t, pa[t], keep_a[t], u % (CIT / 4), u / (CIT / 4));
        }
#pragma unroll
        for (int t = 0; t < NG; ++t) {
            const int u = tid + 512 * t;
            if (u < UG) put(g_t, pg[t], keep_g[t], u % (COT / 4), u / (COT / 4));
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
#pragma unroll
        for (int s = 0; s < kDwPairs / 32; ++s) {         // MFMA step: pairs 32 s + 8 (lane >> 4) .. + 7
            const char* ar = a_t + li * PB + 64 * s + 16 * lq;
            const char* gr = g_t + (16 * CB * wave + li) * PB + 64 * s + 16 * lq;
            bf16x8 bf[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) bf[c] = *reinterpret_cast<const bf16x8*>(gr + 16 * c * PB);
#pragma unroll
            for (int b = 0; b < NBI; ++b) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(ar + 16 * b * PB);
#pragma unroll
                for (int c = 0; c < CB; ++c) acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[c], acc[b][c], 0, 0, 0);
            }
        }
    }
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

template <int NBI, int CB, bool IDENT, bool ABF>
static int launch_bwd_w_bf16(const float* in_a, int c_in_a, const float* in_b, int c_in_b, const float* g,
                             const int32_t* pin, const int32_t* pout, const int32_t* off, int k_vol, int64_t m_out,
                             int64_t n_pairs, int c_out, float* dw, float* workspace, hipStream_t st) {
    constexpr int CIT = 16 * NBI, COT = 128 * CB;
    const size_t lds = (size_t)(CIT + COT) * (2 * kDwPairs + 16);
    auto kern = spconv_bwd_w_bf16_kernel<NBI, CB, IDENT, ABF>;
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

using namespace lidiff;

extern "C" int lidiff_spconv_bwd_w_bf16(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                        const float* grad_out, const int32_t* pairs_in, const int32_t* pairs_out,
                                        const int32_t* offset_ptr, int64_t n_pairs, int32_t k_vol, int64_t m_in,
                                        int64_t m_out, int32_t c_out, float* dw, float* workspace, int32_t in_bf16, void* stream) {
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
    const int nbi = c_in > 128 ? 16 : c_in > 64 ? 8 : c_in > 32 ? 4 : 2;      // as lidiff_spconv_bwd_w (same workspace size)
#define LIDIFF_DW_(NBI, CB, ID, AB)                                                                                      \
    launch_bwd_w_bf16<NBI, CB, ID, AB>(in_a, c_in_a, in_b, c_in_b, grad_out, pairs_in, pairs_out, offset_ptr, k_vol, m_out,  \
                                       n_pairs, c_out, dw, workspace, st)
#define LIDIFF_DW(NBI, CB)                                                                                               \
    return identity ? (in_bf16 ? LIDIFF_DW_(NBI, CB, true, true) : LIDIFF_DW_(NBI, CB, true, false))                     \
                    : (in_bf16 ? LIDIFF_DW_(NBI, CB, false, true) : LIDIFF_DW_(NBI, CB, false, false))
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
#undef LIDIFF_DW_
}

// ---------------------------------------------------------------------------------------
// fp32 -> bf16 rows (round to nearest even): the shadow copy of a feature matrix the bf16 convolutions gather from
__global__ void cast_bf16_kernel(const float* __restrict__ src, int64_t n8, int64_t n, __bf16* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n8) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        const lidiff::f32x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        reinterpret_cast<lidiff::bf16x8*>(dst)[i] = __builtin_convertvector(v, lidiff::bf16x8);
    } else if (i == n8) {
        for (int64_t e = 8 * n8; e < n; ++e) dst[e] = (__bf16)src[e];
    }
}

extern "C" int lidiff_cast_bf16(const float* src, int64_t n, void* dst, void* stream) {
    LIDIFF_CHECK_ARG(n >= 0, "negative size");
    if (n == 0) return 0;
    LIDIFF_CHECK_ARG(src != nullptr && dst != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pointers must be 16-byte aligned");
    const int64_t n8 = n / 8;
    cast_bf16_kernel<<<(unsigned)ceil_div(n8 + 1, 256), 256, 0, (hipStream_t)stream>>>(src, n8, n, (__bf16*)dst);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}


extern "C" int64_t lidiff_spconv_packed_weight_bf16_elems(int32_t k_vol, int32_t c_in, int32_t c_out, int32_t planes) {
    return (int64_t)k_vol * ((c_in + 31) / 32) * 32 * c_out * planes;
}

extern "C" int lidiff_spconv_pack_weights_bf16(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out, int32_t planes,
                                               void* w_packed, void* stream) {
    LIDIFF_CHECK_ARG(w != nullptr && w_packed != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && c_in > 0, "kernel volume must be 1..27, c_in > 0");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 16 == 0, "c_out must be a multiple of 16");
    LIDIFF_CHECK_ARG(planes >= 1 && planes <= 3, "planes must be 1, 2 or 3");
    const int nslab = (c_in + 31) / 32;
    const int64_t total = lidiff_spconv_packed_weight_bf16_elems(k_vol, c_in, c_out, planes);
    pack_weights_bf16_kernel<<<(unsigned)ceil_div(total, 256), 256, 0, (hipStream_t)stream>>>(w, k_vol, c_in, c_out, nslab, planes,
                                                                                             (__bf16*)w_packed, total);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

extern "C" int lidiff_spconv_fwd_bf16(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b, const void* w_packed,
                                      int32_t planes, const int32_t* nbr, int32_t k_vol, int64_t m_in, int64_t m_out,
                                      int32_t c_out, float* out, const float* ep_scale, const float* ep_shift,
                                      const float* residual, int32_t relu, int32_t replicas, int32_t in_bf16, void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0 && w_packed != nullptr && out != nullptr, "null pointer");
    LIDIFF_CHECK_ARG(in_bf16 >= 0 && in_bf16 <= 4, "bf16 feature rows: in_bf16 in 0..4");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(planes == 1, "planes must be 1 (operands rounded to bf16: the training convolution); fp32-accurate results from "
                                  "split operands: lidiff_spconv_fwd_split3");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(nbr != nullptr || (k_vol == 1 && m_in == m_out), "identity map needs K=1, m_in==m_out");
    LIDIFF_CHECK_ARG(c_in_a % 32 == 0 && c_in_b % 32 == 0, "the bf16 kernel needs input widths that are multiples of 32");
    LIDIFF_CHECK_ARG(c_out % 32 == 0, "the bf16 kernel needs c_out % 32 == 0");
    LIDIFF_CHECK_ARG(replicas >= 1 && m_out >= 0 && m_in >= 0, "bad shape");
    if (m_out == 0) return 0;
    LIDIFF_CHECK_ARG(m_in > 0, "outputs without inputs");
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(in_a) && al16(in_b) && al16(w_packed) && al16(out) && al16(ep_scale) && al16(ep_shift) && al16(residual),
                     "pointers must be 16-byte aligned");
    LIDIFF_CHECK_ARG(m_in * (int64_t)c_in_a * 4 < (1ll << 31) && m_in * (int64_t)c_in_b * 4 < (1ll << 31),
                     "a feature matrix exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(lidiff_spconv_packed_weight_bf16_elems(k_vol, c_in_a + c_in_b, c_out, planes) * 2 < (1ll << 31),
                     "packed weights exceed the 2 GiB buffer-descriptor range");
    ConvParams p{};
    p.in_a = in_a; p.in_b = in_b; p.wp = reinterpret_cast<const float*>(w_packed); p.nbr = nbr; p.out = out;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.m_in = m_in; p.m_out = m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu; p.replicas = replicas;
    // measurement aid (tools/conv_probe.py --kernel bf16): LIDIFF_BF16_PROBE bits switch parts of the kernel off -- 1 no
    // gather traffic (zero-filled rows), 2 one hot W slab, 4 no fragment reads / MFMAs, 8 no flush, 16 no gather requests,
    // 32 no stage barrier.  Results are wrong with any bit set.
    static const int probe = [] {
        const char* e = getenv("LIDIFF_BF16_PROBE");
        const int v = e ? atoi(e) : 0;
        if (v) fprintf(stderr, "lidiff_amd: LIDIFF_BF16_PROBE=%d -- parts of lidiff_spconv_fwd_bf16 are switched OFF, its results are WRONG (measurement only)\n", v);
        return v;
    }();
    p.probe = probe;
    hipStream_t st = (hipStream_t)stream;
    const bool ks64 = c_in_a % 64 == 0 && c_in_b % 64 == 0;
    if (in_bf16) return dispatch_bf16_rows(p, ks64, in_bf16, st);
    return dispatch_bf16<1>(p, ks64, st);
}
