// Sparse convolution for gfx950: output-stationary, pair-compacted, fp32 MFMA.
//
// One workgroup (4 waves) owns a tile of BM consecutive OUTPUT rows x BN output channels and
// keeps that tile's fp32 accumulators in LDS for the whole kernel-volume loop, so every
// output row is written to HBM exactly once (with the BatchNorm/ReLU/residual epilogue fused)
// and there are no global atomics -- results are deterministic.
//
// For every kernel offset k the tile's column of the neighbour table nbr[k, row0:row0+BM] is
// compacted with a wave ballot into a dense pair list (input row, local output row).  The
// offset's contribution is then a small dense GEMM
//        [n_k pairs x C_in] (gathered rows)  @  W[k] [C_in x BN]
// executed with v_mfma_f32_32x32x2_f32 in 32-pair row blocks: only ceil(n_k/32) row blocks
// are issued, so MFMA work tracks the REAL pair count (plus <32 rows of padding per offset)
// instead of BM x 27 as a zero-padded implicit GEMM would.  The register accumulators of an
// offset are flushed into the LDS tile through the pair list's local output row (each output
// row occurs at most once per offset, and waves own disjoint row-block/column-block sets, so
// plain LDS read-modify-write is race free).
//
// Data movement per K-slab (KS input channels): gathered A rows (coalesced float4, a full
// 128-byte line per row for KS = 32) and the W[k] slab are prefetched global->registers TWO
// slabs ahead (two register sets) so the gather latency overlaps two slabs of MFMA work, then
// stored to LDS (A rows padded to KS+4 floats: conflict-free ds_read_b128 of 4 consecutive k
// per lane -- K is consumed in a permuted order shared by both operands; B rows are read
// lane-contiguously).  Eight waves per workgroup (two per SIMD) share the MFMA pipe.
//
// Replaces ME's ConvolutionForwardGPU (gather -> GEMM -> atomic scatter per offset) behind
// MinkowskiConvolution / MinkowskiConvolutionTranspose; call sites in include/lidiff_amd.h.
#include <type_traits>

#include "common.h"

namespace lidiff {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (V#) over [p, p+bytes): stride 0, 32-bit data format.  Out-of-range
// offsets return zeros, which is how lanes with nothing to load get their zero fill.
__device__ __forceinline__ u32x4 make_rsrc(const void* p, unsigned bytes) {
    const uint64_t a = (uint64_t)p;
    u32x4 r;
    r.x = (unsigned)a;
    r.y = (unsigned)(a >> 32) & 0xffffu;
    r.z = bytes;
    r.w = 0x00020000u;
    return r;
}

// 16-byte buffer load the COMPILER DOES NOT TRACK: hipcc (ROCm 7.2) waits vmcnt(0) in front of
// the LDS stores of the older register set, which also drains the younger set issued one stage
// ago.  The loads are issued from inline asm and retired by the counted waits below instead
// (cdna_hip_programming.md 5.7: '=v' loads + a wait statement naming every destination).
__device__ __forceinline__ void buffer_load_x4(f32x4& dst, unsigned voff, u32x4 rsrc) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}

struct ConvParams {
    const float* in_a;
    const float* in_b;
    const float* w;
    const int32_t* nbr;
    float* out;
    const float* scale;
    const float* shift;
    const float* residual;
    int64_t m_in, m_out;
    int c_in_a, c_in_b, c_in, c_out, k_vol, relu;
    int tiles_m, tiles_n;
    long long* dbg;      // LIDIFF_CONV_TIMING builds only: per-wave phase cycle sums
};

#ifdef LIDIFF_CONV_TIMING
#define TSTAMP(x) const long long x = __builtin_readcyclecounter()
#define TADD(slot, a, b) tsum[slot] += (b) - (a)
#else
#define TSTAMP(x)
#define TADD(slot, a, b)
#endif

template <int BM, int BN, int KS>
struct ConvCfg {
    static constexpr int kThreads = 512;                 // 8 waves: two per SIMD share the MFMA pipe
    static constexpr int kWaves = kThreads / 64;
    static constexpr int NCB = BN / 32;                  // 32-col blocks in the tile
    static constexpr int NRB = BM / 32;                  // 32-pair row blocks (upper bound)
    static constexpr int MAXB = (NRB * NCB + kWaves - 1) / kWaves;   // MFMA blocks per wave
    static constexpr int LDA = KS + 4;                   // 16-B aligned rows, conflict-free b128 reads
    static constexpr int A_VEC = (BM * KS / 4 + kThreads - 1) / kThreads;   // float4 per thread
    static constexpr int B_VEC = (KS * BN / 4 + kThreads - 1) / kThreads;
    static constexpr int A_SCL = (BM * KS + kThreads - 1) / kThreads;
    static constexpr int B_SCL = (KS * BN + kThreads - 1) / kThreads;
    static_assert(BM % 64 == 0 && BM <= 256, "BM");
    static_assert(MAXB <= 2, "at most two MFMA blocks per wave");
    static_assert(BN % 32 == 0, "BN");
    static_assert(KS % 8 == 0, "KS");

    static constexpr int SLAB = KS * BN + BM * LDA;      // floats per (B slab + A slab) buffer

    __host__ __device__ static size_t lds_bytes(int k_vol) {
        size_t b = (size_t)BM * BN * 4 + 2 * (size_t)SLAB * 4;   // acc tile + double-buffered slabs
        b += (size_t)k_vol * BM * 4;      // in_list
        b += (size_t)64 * 4 * 2;          // cnt, klist (k_vol <= 64)
        b += (size_t)k_vol * BM;          // out_list (uint8)
        b = (b + 15) & ~(size_t)15;
        return b + 64 * 4;                // per-lane dummy words for the branch-free flush
    }
};

template <int BM, int BN, int KS, bool VEC>
__global__ __launch_bounds__(512) void spconv_fwd_kernel(const ConvParams p) {
    using Cfg = ConvCfg<BM, BN, KS>;
    constexpr int LDA = Cfg::LDA;
    constexpr int NT = Cfg::kThreads;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc_lds = reinterpret_cast<float*>(smem);
    float* slab0 = acc_lds + BM * BN;                    // 2 x { Bs [KS][BN], As [BM][LDA] }
    int32_t* in_list = reinterpret_cast<int32_t*>(slab0 + 2 * Cfg::SLAB);
    int32_t* cnt = in_list + p.k_vol * BM;
    int32_t* klist = cnt + 64;
    uint8_t* out_list = reinterpret_cast<uint8_t*>(klist + 64);
    // float index (relative to acc_lds) of 64 dummy words behind everything else
    const int dummy_off = (int)((Cfg::lds_bytes(p.k_vol) - 64 * 4) / 4);

    // XCD-aware tile mapping: the column blocks of one row tile share an XCD (their gathers
    // hit the same L2), consecutive row tiles round-robin over the 8 XCDs.
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform -> scalar branches

    // ---- pair lists: ordered compaction of nbr[k, row0 : row0+rows_here] per offset --------
    if (p.nbr == nullptr) {                      // kernel_size == 1: identity map
        for (int r = tid; r < BM; r += NT) {
            in_list[r] = (int32_t)(row0 + r);
            out_list[r] = (uint8_t)r;
        }
        if (tid == 0) cnt[0] = rows_here;
    } else {
        for (int k = wave; k < p.k_vol; k += Cfg::kWaves) {
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
    // ---- zero the accumulator tile -----------------------------------------------------------
    for (int e = tid; e < BM * BN / 4; e += NT)
        reinterpret_cast<float4*>(acc_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    int nact;
    {   // active offsets (n_k > 0), in ascending k: every wave computes the same list
        const bool act = lane < p.k_vol && cnt[lane] > 0;
        const unsigned long long m = __ballot(act);
        if (wave == 0 && act) klist[popc_below(m)] = lane;
        nact = __popcll(m);
    }
    __syncthreads();

    const int nslab = (p.c_in + KS - 1) / KS;
    const int nit = nact * nslab;

    struct Regs {
        f32x4 a_v[VEC ? Cfg::A_VEC : 1];
        f32x4 b_v[VEC ? Cfg::B_VEC : 1];
        float a_s[VEC ? 1 : Cfg::A_SCL];
        float b_s[VEC ? 1 : Cfg::B_SCL];
    };

    // buffer descriptors (built from kernel arguments only: wave-uniform, live in SGPRs)
    const u32x4 rsrc_a = make_rsrc(p.in_a, (unsigned)(p.m_in * p.c_in_a * 4));
    const u32x4 rsrc_b = make_rsrc(p.in_b ? p.in_b : p.in_a, (unsigned)(p.m_in * p.c_in_b * 4));
    const u32x4 rsrc_w = make_rsrc(p.w, (unsigned)(p.k_vol * p.c_in * p.c_out * 4));

    // counted retirement of the asm loads: wait until at most `keep` vector-memory operations are
    // outstanding; naming every register of the set keeps the compiler from touching them earlier
    auto retire = [&](Regs& rg) {
        if constexpr (VEC) {
            constexpr int NL = Cfg::A_VEC + Cfg::B_VEC;      // loads of the younger set stay in flight
            static_assert(Cfg::A_VEC == 2 && (Cfg::B_VEC == 1 || Cfg::B_VEC == 2), "register-set shape");
            if constexpr (Cfg::B_VEC == 2)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rg.a_v[0]), "+v"(rg.a_v[1]), "+v"(rg.b_v[0]), "+v"(rg.b_v[1]) : "n"(NL) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(%3)" : "+v"(rg.a_v[0]), "+v"(rg.a_v[1]), "+v"(rg.b_v[0]) : "n"(NL) : "memory");
        }
    };

    // Position in the (active offset, K-slab) sequence, kept in SGPRs and advanced incrementally
    // (no integer division, one LDS lookup per OFFSET rather than per slab).
    struct Cursor { int ai, slab, k, nk; };
    auto cursor_load = [&](Cursor& c) {
        if (c.ai < nact) {
            c.k = __builtin_amdgcn_readfirstlane(klist[c.ai]);
            c.nk = __builtin_amdgcn_readfirstlane(cnt[c.k]);
        } else {
            c.k = 0;
            c.nk = 0;                          // past the end: every lane loads nothing
        }
    };
    auto cursor_next = [&](Cursor& c) {
        if (++c.slab == nslab) {
            c.slab = 0;
            ++c.ai;
            cursor_load(c);
        }
    };
    Cursor pf{0, 0, 0, 0}, cur{0, 0, 0, 0};
    cursor_load(pf);
    cursor_load(cur);

    // global -> registers for the slab at the prefetch cursor (zero-filled beyond n_k rows / c_in
    // channels), then advance the cursor.  Past the end the same loads are issued with
    // out-of-range offsets (zeros, no traffic): every stage then has exactly one younger register
    // set in flight and ONE counted wait fits all.
    auto prefetch = [&](Regs& rg) {
        const bool live = pf.ai < nact;
        const int k = pf.k;
        const int k0 = pf.slab * KS;
        const int n_k = pf.nk;
        if constexpr (VEC) {
            // Buffer loads through wave-uniform descriptors: a lane that has nothing to load
            // (row >= n_k, channel >= c_in) passes an out-of-range offset and the hardware
            // bounds check returns zeros -- no branches around the loads.
            const bool from_a = k0 < p.c_in_a;             // uniform: slabs never straddle a|b
            const u32x4 rs = from_a ? rsrc_a : rsrc_b;
            const int cw = from_a ? p.c_in_a : p.c_in_b;
            const int cbase = from_a ? k0 : k0 - p.c_in_a;
            unsigned rowv[Cfg::A_VEC];
#pragma unroll
            for (int j = 0; j < Cfg::A_VEC; ++j)        // both list reads in flight, one LDS wait
                rowv[j] = (unsigned)in_list[k * BM + min((tid + j * NT) / (KS / 4), BM - 1)];
#pragma unroll
            for (int j = 0; j < Cfg::A_VEC; ++j) {
                const int e = tid + j * NT;
                const int pos = e / (KS / 4), cl = 4 * (e % (KS / 4));
                bool ok = pos < n_k && k0 + cl < p.c_in;
                if constexpr ((Cfg::A_VEC) * NT > BM * KS / 4) ok = ok && e < BM * KS / 4;
                const unsigned off = ok ? (rowv[j] * (unsigned)cw + (unsigned)(cbase + cl)) * 4u : 0xFFFFFFF0u;
                buffer_load_x4(rg.a_v[j], off, rs);
            }
#pragma unroll
            for (int j = 0; j < Cfg::B_VEC; ++j) {
                const int e = tid + j * NT;
                const int kr = e / (BN / 4), cq = e % (BN / 4);
                bool ok = live && k0 + kr < p.c_in;
                if constexpr ((Cfg::B_VEC) * NT > KS * BN / 4) ok = ok && e < KS * BN / 4;
                const unsigned off = ok ? (((unsigned)k * p.c_in + k0 + kr) * (unsigned)p.c_out + n0 + 4 * cq) * 4u
                                        : 0xFFFFFFF0u;
                buffer_load_x4(rg.b_v[j], off, rsrc_w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < Cfg::A_SCL; ++j) {
                const int e = tid + j * NT;
                const int pos = e / KS, col = k0 + e % KS;
                float v = 0.f;
                if (live && e < BM * KS && pos < n_k && col < p.c_in) {
                    const int64_t row = in_list[k * BM + pos];
                    v = (col < p.c_in_a) ? p.in_a[row * p.c_in_a + col]
                                         : p.in_b[row * p.c_in_b + (col - p.c_in_a)];
                }
                rg.a_s[j] = v;
            }
#pragma unroll
            for (int j = 0; j < Cfg::B_SCL; ++j) {
                const int e = tid + j * NT;
                const int kr = e / BN, cc = e % BN;
                float v = 0.f;
                if (live && e < KS * BN && k0 + kr < p.c_in)
                    v = p.w[((int64_t)k * p.c_in + k0 + kr) * p.c_out + n0 + cc];
                rg.b_s[j] = v;
            }
        }
        cursor_next(pf);
    };

    auto store_slab = [&](const Regs& rg, int buf) {
        float* Bs = slab0 + buf * Cfg::SLAB;
        float* As = Bs + KS * BN;
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < Cfg::A_VEC; ++j) {
                const int e = tid + j * NT;
                if ((j + 1) * NT <= BM * KS / 4 || e < BM * KS / 4)
                    *reinterpret_cast<f32x4*>(As + (e / (KS / 4)) * LDA + 4 * (e % (KS / 4))) = rg.a_v[j];
            }
#pragma unroll
            for (int j = 0; j < Cfg::B_VEC; ++j) {
                const int e = tid + j * NT;
                if ((j + 1) * NT <= KS * BN / 4 || e < KS * BN / 4) reinterpret_cast<f32x4*>(Bs)[e] = rg.b_v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < Cfg::A_SCL; ++j) {
                const int e = tid + j * NT;
                if (e < BM * KS) As[(e / KS) * LDA + e % KS] = rg.a_s[j];
            }
#pragma unroll
            for (int j = 0; j < Cfg::B_SCL; ++j) {
                const int e = tid + j * NT;
                if (e < KS * BN) Bs[e] = rg.b_s[j];
            }
        }
    };

    const int l31 = lane & 31, lhi = lane >> 5;

#ifdef LIDIFF_CONV_TIMING
    long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    floatx16 acc[Cfg::MAXB];
#pragma unroll
    for (int s = 0; s < Cfg::MAXB; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    // One pipeline stage.  On entry LDS buffer (it & 1) holds slab `it` (published by the
    // previous barrier) and `rg` holds slab it+1, loaded two stages ago.  The stage writes slab
    // it+1 into the OTHER buffer, refills `rg` with slab it+3, multiplies slab `it`, and ends with
    // the single barrier of the iteration -- LDS stores and global loads issue in the shadow of
    // the 64-cycle MFMAs.  MFMA blocks (row block, col block) of the offset are dealt round-robin
    // to the 8 waves.  K is consumed in a permuted order: step (j, e) takes
    // k = 8j + 4*(lane>>5) + e from BOTH operands, so an A fragment is one ds_read_b128 per 4 MFMAs.
    auto stage = [&](int it, Regs& rg) {
        const float* Bs = slab0 + (it & 1) * Cfg::SLAB;
        const float* As = Bs + KS * BN;
        const int k = cur.k;
        const int slab = cur.slab;
        const int n_k = cur.nk;
        const int nrb = (n_k + 31) >> 5;
        const int nblk = nrb * Cfg::NCB;

        // staging half of the stage: slab it+1 registers -> the other LDS buffer, refill with it+3
        auto stage_io = [&]() {
            TSTAMP(t0);
            retire(rg);                       // the other set (slab it+2, maybe a dummy) stays in flight
            TSTAMP(t1);
            store_slab(rg, (it + 1) & 1);     // past the end: zeros into the free buffer, harmless
            TSTAMP(t2);
            prefetch(rg);
            TSTAMP(t3);
            TADD(0, t0, t1); TADD(1, t1, t2); TADD(2, t2, t3);
        };

        // MFMA half: this wave's active blocks are s = 0 .. nb_w-1 (scalar).  Fragments are
        // double-buffered in registers: the ds_reads of step j+1 are issued before the MFMAs of
        // step j (sched_barrier keeps them there), so LDS latency hides under 4*NB MFMAs.
        const int nb_w = (nblk > wave) ? (nblk - wave + Cfg::kWaves - 1) / Cfg::kWaves : 0;
        auto mma = [&](auto nb_tag) {
            constexpr int NB = decltype(nb_tag)::value;
            const float* ap[NB];
            const float* bp[NB];
#pragma unroll
            for (int s = 0; s < NB; ++s) {
                const int b = wave + s * Cfg::kWaves;
                const int rb = b / Cfg::NCB, cb = b % Cfg::NCB;
                ap[s] = As + (rb * 32 + l31) * LDA + 4 * lhi;
                bp[s] = Bs + (4 * lhi) * BN + cb * 32 + l31;
            }
            struct Frag { float4 a[NB]; float b[NB][4]; };
            auto load_frag = [&](int j, Frag& f) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    f.a[s] = *reinterpret_cast<const float4*>(ap[s] + 8 * j);
                    const float* q = bp[s] + 8 * j * BN;
                    f.b[s][0] = q[0]; f.b[s][1] = q[BN]; f.b[s][2] = q[2 * BN]; f.b[s][3] = q[3 * BN];
                }
            };
            auto issue = [&](const Frag& f) {
#pragma unroll
                for (int s = 0; s < NB; ++s) {
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s].x, f.b[s][0], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s].y, f.b[s][1], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s].z, f.b[s][2], acc[s], 0, 0, 0);
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s].w, f.b[s][3], acc[s], 0, 0, 0);
                }
            };
            static_assert(KS / 8 == 4, "fragment pipeline written for 4 steps per slab");
            Frag f0, f1;
            load_frag(0, f0);
            load_frag(1, f1);
            __builtin_amdgcn_sched_barrier(0);
            issue(f0);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(2, f0);
            __builtin_amdgcn_sched_barrier(0);
            issue(f1);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(3, f1);
            __builtin_amdgcn_sched_barrier(0);
            issue(f0);
            __builtin_amdgcn_sched_barrier(0);
            issue(f1);
        };
        auto mma_any = [&]() {
            if constexpr (Cfg::MAXB >= 2) {
                if (nb_w >= 2) mma(std::integral_constant<int, 2>{});
                else if (nb_w == 1) mma(std::integral_constant<int, 1>{});
            } else {
                if (nb_w >= 1) mma(std::integral_constant<int, 1>{});
            }
        };

        // The two waves of a SIMD (w and w+4) run the two halves in OPPOSITE order, so one
        // wave's staging (LDS stores, address generation, global loads) overlaps the other
        // wave's MFMAs instead of both leaving the matrix pipe idle at the same time.
        // (stage_io appears ONCE in the instruction stream: its asm-loaded registers must not pass
        // through a control-flow merge, or the compiler inserts copies ahead of the counted wait.)
        const bool io_first = wave < Cfg::kWaves / 2;
        TSTAMP(m0);
        if (!io_first) mma_any();
        TSTAMP(m1);
        stage_io();
        TSTAMP(m2);
        if (io_first) mma_any();
        TSTAMP(m3);
        TADD(3, m0, m1); TADD(3, m2, m3);

        if (slab == nslab - 1) {         // offset finished: flush registers into the LDS tile
            // Batched and branch-free: 16 list lookups, then 16 tile reads, then 16 writes per
            // block (3 LDS round trips instead of 48); rows beyond n_k go to a per-lane dummy word.
#pragma unroll
            for (int s = 0; s < Cfg::MAXB; ++s) {
                const int b = wave + s * Cfg::kWaves;
                if (b < nblk) {
                    const int rb = b / Cfg::NCB, cb = b % Cfg::NCB;
                    const int col = cb * 32 + l31;
                    int addr[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int prow = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const int orow = out_list[k * BM + prow];
                        addr[r] = prow < n_k ? orow * BN + col : dummy_off + lane;
                    }
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) old[r] = acc_lds[addr[r]];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc_lds[addr[r]] = old[r] + acc[s][r];
                        acc[s][r] = 0.f;
                    }
                }
            }
        }
        cursor_next(cur);
        TSTAMP(b0);
        __syncthreads();                 // slab it+1 visible, buffer (it & 1) free again
        TSTAMP(b1);
        TADD(4, m3, b0); TADD(5, b0, b1); TADD(6, m0, b1);
    };

#ifdef LIDIFF_CONV_TIMING
    const long long t_begin = __builtin_readcyclecounter();
#endif
    Regs r0, r1;
    if (nit > 0) {
        prefetch(r0);
        prefetch(r1);
        retire(r0);
        store_slab(r0, 0);
        prefetch(r0);
    }
    __syncthreads();
    for (int it = 0; it < nit; it += 2) {     // stage(it) consumes the register set holding slab it+1
        stage(it, r1);
        if (it + 1 < nit) stage(it + 1, r0);
    }
    if constexpr (VEC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing dummy loads
#ifdef LIDIFF_CONV_TIMING
    if (p.dbg != nullptr && lane == 0) {
        tsum[7] = __builtin_readcyclecounter() - t_begin;      // whole slab loop incl. pipeline fill
        long long* d = p.dbg + ((long long)blockIdx.x * Cfg::kWaves + wave) * 9;
        for (int q = 0; q < 8; ++q) d[q] = tsum[q];
        d[8] = nit;
    }
#endif
    __syncthreads();

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

template <int BM, int BN, int KS, bool VEC>
static int launch_fwd(const ConvParams& p, hipStream_t st) {
    using Cfg = ConvCfg<BM, BN, KS>;
    const size_t lds = Cfg::lds_bytes(p.k_vol);
    LIDIFF_CHECK_ARG(lds <= 160 * 1024, "LDS budget exceeded");
    auto kern = spconv_fwd_kernel<BM, BN, KS, VEC>;
    static thread_local size_t configured = 0;
    if (lds > configured) {
        LIDIFF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    ConvParams q = p;
    q.tiles_m = (int)ceil_div(p.m_out, BM);
    q.tiles_n = p.c_out / BN;
    const unsigned grid = (unsigned)(ceil_div(q.tiles_m, 8) * 8 * q.tiles_n);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::kThreads), lds, st, q);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

template <int BN>
static int dispatch_fwd(const ConvParams& p, bool vec, hipStream_t st) {
    if (vec) return launch_fwd<128, BN, 32, true>(p, st);
    return launch_fwd<128, BN, 32, false>(p, st);
}

}  // namespace lidiff

using namespace lidiff;

static long long* g_conv_dbg = nullptr;
#ifdef LIDIFF_CONV_TIMING
extern "C" void lidiff_debug_set_conv_timing_buffer(long long* d_buf) { g_conv_dbg = d_buf; }
#endif

extern "C" int lidiff_spconv_fwd(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                                 const float* w, const int32_t* nbr, int32_t k_vol, int64_t m_in,
                                 int64_t m_out, int32_t c_out, float* out, const float* ep_scale,
                                 const float* ep_shift, const float* residual, int32_t relu,
                                 void* stream) {
    LIDIFF_CHECK_ARG(in_a != nullptr && c_in_a > 0, "in_a / c_in_a");
    LIDIFF_CHECK_ARG((in_b == nullptr) == (c_in_b == 0), "in_b and c_in_b must agree");
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27, "kernel volume must be 1..27");
    LIDIFF_CHECK_ARG(nbr != nullptr || (k_vol == 1 && m_in == m_out), "identity map needs K=1, m_in==m_out");
    LIDIFF_CHECK_ARG(c_out > 0 && c_out % 32 == 0, "c_out must be a multiple of 32");
    LIDIFF_CHECK_ARG(m_out >= 0 && m_in >= 0, "negative rows");
    if (m_out == 0) return 0;
    ConvParams p{};
    p.in_a = in_a; p.in_b = in_b; p.w = w; p.nbr = nbr; p.out = out;
    p.scale = ep_scale; p.shift = ep_shift; p.residual = residual;
    p.m_in = m_in; p.m_out = m_out;
    p.c_in_a = c_in_a; p.c_in_b = c_in_b; p.c_in = c_in_a + c_in_b; p.c_out = c_out;
    p.k_vol = k_vol; p.relu = relu; p.dbg = g_conv_dbg;
    auto al16 = [](const void* q) { return q == nullptr || ((uintptr_t)q & 15) == 0; };
    LIDIFF_CHECK_ARG(al16(w) && al16(out) && al16(ep_scale) && al16(ep_shift) && al16(residual),
                     "w/out/epilogue pointers must be 16-byte aligned");
    const bool fits32 = m_in * (int64_t)c_in_a * 4 < (1ll << 31) && m_in * (int64_t)c_in_b * 4 < (1ll << 31) &&
                        (int64_t)k_vol * (c_in_a + c_in_b) * c_out * 4 < (1ll << 31);
    LIDIFF_CHECK_ARG(fits32, "a feature or weight matrix exceeds the 2 GiB buffer-descriptor range");
    LIDIFF_CHECK_ARG(c_in_b == 0 || c_in_a % 32 == 0, "with two inputs c_in_a must be a multiple of 32 (slab size)");
    const bool vec = c_in_a % 4 == 0 && c_in_b % 4 == 0 && al16(in_a) && al16(in_b);
    hipStream_t st = (hipStream_t)stream;
    if (c_out % 128 == 0) return dispatch_fwd<128>(p, vec, st);
    if (c_out % 96 == 0) return dispatch_fwd<96>(p, vec, st);
    if (c_out % 64 == 0) return dispatch_fwd<64>(p, vec, st);
    return dispatch_fwd<32>(p, vec, st);
}

extern "C" int lidiff_spconv_bwd_w(const float*, int32_t, const float*, int32_t, const float*,
                                   const int32_t*, int32_t, int64_t, int64_t, int32_t, float*, void*) {
    set_error("lidiff_spconv_bwd_w: not built yet");
    return 3;
}
