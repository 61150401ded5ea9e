// Coordinate-side kernels: voxel hashing, first-occurrence unique, inverse map, voxel mean,
// strided maps, kernel maps (neighbour tables), ME-layout rulebook compaction, row
// gather/scatter, nearest-part-voxel match.  All HBM-bound integer/byte work: one row per
// lane, coalesced row-major traffic, wave ballot + mbcnt for ordered compaction.
// Reference call sites are cited in include/lidiff_amd.h.
#include <stdarg.h>

#include <stdlib.h>

#include "common.h"

namespace lidiff {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

// ---------------------------------------------------------------------------------------
__global__ void floor_kernel(const float* __restrict__ in, int32_t* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)floorf(in[i]);
}

// Key of row i after flooring columns 1..3 to a multiple of s (s == 1: identity).
__device__ __forceinline__ int4 strided_row(const int32_t* __restrict__ coords, int64_t i, int s) {
    int4 c = reinterpret_cast<const int4*>(coords)[i];
    if (s > 1) {
        c.y = floor_div(c.y, s) * s;
        c.z = floor_div(c.z, s) * s;
        c.w = floor_div(c.w, s) * s;
    }
    return c;
}

// Phase 1: insert every row's key; the slot remembers the smallest row index that hit it.
// (every kernel of the chain: d_n != nullptr means the row count lives on the device -- the output of the previous
// level's compaction -- and `n` is only the bound the grid was sized for: no host read between the levels)
__global__ void insert_kernel(const int32_t* __restrict__ coords, int64_t n, const int32_t* __restrict__ d_n, int s,
                              uint64_t* __restrict__ hkeys, int32_t* __restrict__ hvals,
                              uint32_t mask, int32_t* __restrict__ slot_of,
                              int32_t* __restrict__ d_status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = *d_n;
    if (i >= n) return;
    const int4 c = strided_row(coords, i, s);
    bool ok;
    const uint64_t key = pack_key(c.x, c.y, c.z, c.w, ok);
    if (!ok) atomicOr(d_status, LIDIFF_STATUS_KEY_RANGE);
    // Wave-uniform key (x_uncond: every point in ONE voxel; heavy duplicates): only the first
    // active lane -- the smallest row index of the wave -- touches the table, the rest reuse its
    // slot.  Avoids 64-way same-address CAS/min serialisation.
    const uint32_t k_lo = __builtin_amdgcn_readfirstlane((uint32_t)key);
    const uint32_t k_hi = __builtin_amdgcn_readfirstlane((uint32_t)(key >> 32));
    const bool uni = __all(key == (((uint64_t)k_hi << 32) | k_lo));
    const bool do_insert = !uni || (int32_t)i == __builtin_amdgcn_readfirstlane((int32_t)i);
    uint32_t slot = hash_key(key) & mask;
    if (do_insert) {
        bool placed = false;
        for (uint32_t probe = 0; probe <= mask; ++probe) {
            const uint64_t prev = atomicCAS((unsigned long long*)&hkeys[slot],
                                            (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (prev == kEmptyKey || prev == key) { placed = true; break; }
            slot = (slot + 1) & mask;
        }
        if (!placed) { atomicOr(d_status, LIDIFF_STATUS_HASH_FULL); slot = 0; }
        atomicMin(reinterpret_cast<unsigned*>(&hvals[slot]), (unsigned)i);      // (unsigned: an empty slot reads 0xFFFFFFFF, the keys' byte pattern)
    }
    if (uni) slot = __builtin_amdgcn_readfirstlane(slot);
    slot_of[i] = (int32_t)slot;
}

// Phase 2: a row is its voxel's first occurrence iff the slot kept its index.  The flag is
// parked in bit 31 of slot_of; per-block counts feed the ordered compaction.
__global__ void flag_count_kernel(const int32_t* __restrict__ hvals, int64_t n, const int32_t* __restrict__ d_n,
                                  int32_t* __restrict__ slot_of, int32_t* __restrict__ blk_counts) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = *d_n;
    bool first = false;
    if (i < n) {
        const int32_t slot = slot_of[i];
        first = hvals[slot] == (int32_t)i;
        if (first) slot_of[i] = slot | (int32_t)0x80000000;
    }
    const unsigned long long m = __ballot(first);
    if (lane_id() == 0) wave_cnt[threadIdx.x / kWave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += wave_cnt[w];
        blk_counts[blockIdx.x] = t;
    }
}

// Sum of blk_counts[0 .. blockIdx.x) by the whole block.
__device__ __forceinline__ int block_prefix_of_counts(const int32_t* __restrict__ blk_counts,
                                                      int* smem /* kWavesPerBlock ints */) {
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += blockDim.x) part += blk_counts[b];
    for (int off = kWave / 2; off > 0; off >>= 1) part += __shfl_down(part, off);
    if (lane_id() == 0) smem[threadIdx.x / kWave] = part;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) total += smem[w];
    __syncthreads();
    return total;
}

// Phase 3: ordered compaction of the first occurrences: row id = #first occurrences before
// this point.  Writes the unique rows, the first-point index, and the row id into the table.
__global__ void scan_write_kernel(const int32_t* __restrict__ coords, int64_t n, const int32_t* __restrict__ d_n, int s,
                                  const int32_t* __restrict__ slot_of,
                                  const int32_t* __restrict__ blk_counts,
                                  int32_t* __restrict__ hvals, int32_t* __restrict__ uniq,
                                  int32_t* __restrict__ first_idx, int32_t* __restrict__ d_m) {
    __shared__ int smem[kWavesPerBlock];
    __shared__ int wave_cnt[kWavesPerBlock];
    const int base = block_prefix_of_counts(blk_counts, smem);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = *d_n;
    int32_t slot = 0;
    bool first = false;
    if (i < n) {
        slot = slot_of[i];
        first = slot < 0;
        slot &= 0x7fffffff;
    }
    const unsigned long long m = __ballot(first);
    const int w = threadIdx.x / kWave;
    if (lane_id() == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int wave_base = 0, block_total = 0;
    for (int k = 0; k < kWavesPerBlock; ++k) {
        if (k < w) wave_base += wave_cnt[k];
        block_total += wave_cnt[k];
    }
    if (first) {
        const int row = base + wave_base + popc_below(m);
        const int4 c = strided_row(coords, i, s);
        reinterpret_cast<int4*>(uniq)[row] = c;
        if (first_idx) first_idx[row] = (int32_t)i;
        hvals[slot] = row;
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *d_m = base + block_total;
}

// Phase 4: point -> row through the table.
template <typename OutT>
__global__ void inverse_kernel(const int32_t* __restrict__ hvals, const int32_t* __restrict__ slot_of,
                               int64_t n, const int32_t* __restrict__ d_n, OutT* __restrict__ inverse) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_n) n = *d_n;
    if (i < n) inverse[i] = (OutT)hvals[slot_of[i] & 0x7fffffff];
}

static int run_unique(const int32_t* coords, int64_t n, const int32_t* d_n, int s, uint64_t* hkeys, int32_t* hvals,
                      int64_t cap, int32_t* uniq, int32_t* first_idx, void* inverse, bool inverse64,
                      int32_t* d_m, int32_t* d_status, void* workspace, hipStream_t st, bool preinit = false) {
    LIDIFF_CHECK_ARG(n >= 0 && n < (1ll << 30), "row count out of range");
    LIDIFF_CHECK_ARG(cap >= 2 * n && (cap & (cap - 1)) == 0, "cap must be a power of two >= 2*n");
    // keys and values start as 0xFF bytes (the empty key; the largest unsigned row index).  preinit: the caller has filled them --
    // a pyramid's tables and neighbour tables live in ONE pool cleared by ONE memset (ops.build_pyramid_lanes), not one per table
    if (!preinit) {
        LIDIFF_CHECK_HIP(hipMemsetAsync(hkeys, 0xFF, (size_t)cap * sizeof(uint64_t), st));
        LIDIFF_CHECK_HIP(hipMemsetAsync(hvals, 0xFF, (size_t)cap * sizeof(int32_t), st));
    }
    if (n == 0) {
        LIDIFF_CHECK_HIP(hipMemsetAsync(d_m, 0, sizeof(int32_t), st));
        return 0;
    }
    const int nblk = (int)ceil_div(n, kBlock);
    int32_t* slot_of = (int32_t*)workspace;
    int32_t* blk_counts = slot_of + n;
    const uint32_t mask = (uint32_t)(cap - 1);
    insert_kernel<<<nblk, kBlock, 0, st>>>(coords, n, d_n, s, hkeys, hvals, mask, slot_of, d_status);
    flag_count_kernel<<<nblk, kBlock, 0, st>>>(hvals, n, d_n, slot_of, blk_counts);
    scan_write_kernel<<<nblk, kBlock, 0, st>>>(coords, n, d_n, s, slot_of, blk_counts, hvals, uniq,
                                               first_idx, d_m);
    if (inverse64)
        inverse_kernel<int64_t><<<nblk, kBlock, 0, st>>>(hvals, slot_of, n, d_n, (int64_t*)inverse);
    else
        inverse_kernel<int32_t><<<nblk, kBlock, 0, st>>>(hvals, slot_of, n, d_n, (int32_t*)inverse);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------
// voxel mean -- deterministic.  The points of a voxel are summed in 64-bit FIXED POINT with integer atomics: integer
// addition is associative, so the sum does not depend on the order in which the atomics land (fp32 atomics, as ME's GPU path
// uses, made two runs of the same step differ in the last bits).  The scale is 2^shift with shift chosen per call from
// max |x| and the point count so that the sum of all N values cannot overflow 62 bits; with LiDiff's features (metres,
// |x| < 2^7, N < 2^18) shift = 37: every fp32 value above 2^-14 m is represented exactly, the sum is exact, and it is rounded
// to fp32 ONCE -- for one or two points per voxel (x_t: ~1.0 per voxel) that is bit for bit the sequential fp32 sum.
__global__ void mean_absmax_kernel(const float* __restrict__ feats, int64_t total, uint32_t* __restrict__ amax) {
    // ONE atomic per workgroup (round 4: one per wave from 1 024 workgroups were 4 096 serialised atomics on one word -- 50 us on
    // the critical path between two denoising steps for 2 MB of input)
    __shared__ uint32_t wmax[kBlock / kWave];
    uint32_t m = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t b = __float_as_uint(feats[e]) & 0x7fffffffu;   // |x| as bits: monotonic for finite values
        if (b < 0x7f800000u) m = max(m, b);                           // (Inf / NaN take no part in the scale: see mean_accum_kernel)
    }
    for (int off = kWave / 2; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off));
    if (lane_id() == 0) wmax[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / kWave); ++w) m = max(m, wmax[w]);
        if (m) atomicMax(amax, m);
    }
}

// 2^shift: the largest power of two for which n_bits-many values below 2^(exponent of amax + 1) sum to less than 2^62
__device__ __forceinline__ int mean_shift(uint32_t amax_bits, int n_bits) {
    if (amax_bits == 0) return 0;
    const int e = (int)(amax_bits >> 23) - 126;                       // max |x| < 2^e
    return 62 - e - n_bits;
}

constexpr int32_t kMeanPoison = 1 << 30;      // bit 30 of a voxel's point count: one of its features is Inf / NaN

__global__ void mean_accum_kernel(const float* __restrict__ feats, const int64_t* __restrict__ inverse,
                                  int64_t n, int c, int n_bits, const uint32_t* __restrict__ amax,
                                  unsigned long long* __restrict__ acc, int32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < n;
    const int row = valid ? (int)inverse[i] : -1;
    const double scale = ldexp(1.0, mean_shift(*amax, n_bits));
    // An Inf / NaN feature has no fixed-point image: it adds nothing to the sums and marks ITS voxel (bit 30 of the count, N <
    // 2^30), whose channels mean_div_kernel then sums in fp32 in point order -- the non-finite value reaches the voxel it belongs
    // to and no other, as the reference's fp32 sum does (round 5 made the whole batch NaN).
    auto fixed = [&](float x) {
        return (__float_as_uint(x) & 0x7fffffffu) < 0x7f800000u ? __double2ll_rn((double)x * scale) : 0ll;
    };
    if (valid) {
        bool bad = false;
        for (int j = 0; j < c; ++j) bad |= (__float_as_uint(feats[i * c + j]) & 0x7fffffffu) >= 0x7f800000u;
        if (bad) atomicOr(&cnt[row], kMeanPoison);
    }
    // all valid lanes of the wave fall in one voxel (x_uncond, heavy duplicates): reduce in the
    // wave and issue ONE atomic per channel instead of 64 serialised same-address atomics.
    const unsigned long long vm = __ballot(valid);
    if (vm == 0) return;
    const int r0 = __shfl(row, __ffsll((long long)vm) - 1);
    if (__all(!valid || row == r0)) {
        for (int j = 0; j < c; ++j) {
            long long v = valid ? fixed(feats[i * c + j]) : 0ll;
            for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane_id() == 0) atomicAdd(&acc[(int64_t)r0 * c + j], (unsigned long long)v);
        }
        if (lane_id() == 0) atomicAdd(&cnt[r0], (int32_t)__popcll(vm));
        return;
    }
    if (!valid) return;
    for (int j = 0; j < c; ++j)
        atomicAdd(&acc[(int64_t)row * c + j], (unsigned long long)fixed(feats[i * c + j]));
    atomicAdd(&cnt[row], 1);
}

__global__ void mean_div_kernel(const unsigned long long* __restrict__ acc, const int32_t* __restrict__ cnt,
                                int64_t total, int c, int n_bits, const uint32_t* __restrict__ amax,
                                const float* __restrict__ feats, const int64_t* __restrict__ inverse, int64_t n_rows,
                                float* __restrict__ out, float* __restrict__ counts) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const double inv_scale = ldexp(1.0, -mean_shift(*amax, n_bits));
    const int32_t cw = cnt[e / c];
    const float n = (float)(cw & (kMeanPoison - 1));
    float sum = (float)((double)(long long)acc[e] * inv_scale);        // the exact sum, rounded to fp32 once
    if (cw & kMeanPoison) {
        // a voxel holding an Inf / NaN feature: its channels as the plain fp32 sum of its points in point order (a scan of the
        // whole inverse map per element -- the pathological case only), so that the value propagates the way fp32 addition
        // propagates it (NaN; +-Inf; Inf - Inf = NaN), to this voxel and to no other
        // (a channel of that voxel without such a value keeps its exact sum)
        float seq = 0.f;
        bool bad = false;
        for (int64_t i = 0; i < n_rows; ++i)
            if (inverse[i] == e / c) {
                const float x = feats[i * c + e % c];
                bad |= (__float_as_uint(x) & 0x7fffffffu) >= 0x7f800000u;
                seq += x;
            }
        if (bad) sum = seq;
    }
    out[e] = sum / n;
    if (e % c == 0) counts[e / c] = n;
}

__global__ void mean_bwd_kernel(const float* __restrict__ grad_out, const int64_t* __restrict__ inverse,
                                const float* __restrict__ counts, int64_t total, int c,
                                float* __restrict__ grad_feats) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int64_t row = inverse[e / c];
    grad_feats[e] = grad_out[row * c + e % c] / counts[row];
}

// ---------------------------------------------------------------------------------------
// kernel maps
// One output row per lane, all KS^3 offsets of the row in flight together: phase 1 issues the first probe of every offset
// (KS^3 independent 8-byte loads per lane), phase 2 the row-id loads of the hits; only an offset whose first slot holds
// another key (the table is at most half full) walks on sequentially.  Measured on the 180k-row stride-1 map: 87 us
// against 98 us for one hash_find after the other -- and 118 us when phase 1 also fetches the second slot of every offset:
// the kernel is bound by the NUMBER of scattered 8-byte requests (~90 G/s chip-wide), not by the length of the dependent
// chain, so fewer distinct cache lines per row is what would help (a spatially coherent hash), not more parallelism.
template <int KS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 3))) void kernel_map_kernel(
    const int32_t* __restrict__ out_coords, int64_t m_out, const uint64_t* __restrict__ hkeys,
    const int32_t* __restrict__ hvals, uint32_t mask, int step, int32_t* __restrict__ nbr) {
    constexpr int KV = KS * KS * KS;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= m_out) return;
    const int4 c = reinterpret_cast<const int4*>(out_coords)[o];
    constexpr int lo = (KS & 1) ? -(KS - 1) / 2 : 0;
    auto key_of = [&](int k, bool& ok) {
        const int dx = k % KS, dy = (k / KS) % KS, dz = k / (KS * KS);
        return pack_key(c.x, c.y + (dx + lo) * step, c.z + (dy + lo) * step, c.w + (dz + lo) * step, ok);
    };
    uint64_t got[KV];
    uint32_t slot[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {                         // phase 1: the first probe of every offset, no branches
        bool ok;
        slot[k] = hash_key(key_of(k, ok)) & mask;
        got[k] = hkeys[slot[k]];
    }
    uint32_t hit = 0, walk = 0;
    int val[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) {                         // phase 2: row ids of the hits (misses read slot 0: one cached line)
        bool ok;
        const uint64_t key = key_of(k, ok);
        const bool h = ok && got[k] == key;
        hit |= h ? 1u << k : 0u;
        walk |= (ok && !h && got[k] != kEmptyKey) ? 1u << k : 0u;
        val[k] = hvals[slot[k] & (0u - (uint32_t)h)];
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) asm volatile("" ::"v"(val[k]));      // all row-id loads issued here, not sunk into the branches below
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        int res = (hit >> k) & 1u ? val[k] : -1;
        if ((walk >> k) & 1u) {                            // first slot taken by another key: walk on
            bool ok;
            const uint64_t key = key_of(k, ok);
            uint32_t sl = (slot[k] + 1) & mask;
            for (uint32_t probe = 1; probe <= mask; ++probe) {
                const uint64_t q = hkeys[sl];
                if (q == key) { res = hvals[sl]; break; }
                if (q == kEmptyKey) break;
                sl = (sl + 1) & mask;
            }
        }
        nbr[(int64_t)k * m_out + o] = res;
    }
}

// kernel_size-3 map of a coordinate map ONTO ITSELF: row o is the neighbour of row i under offset k exactly when i is the
// neighbour of o under the mirrored offset 26 - k, and the centre offset is the identity.  So only the 13 offsets below the
// centre are looked up; every hit also writes its mirror entry (each table slot is written by at most one row: no race, the
// table equals kernel_map_kernel<3>'s bit for bit).  nbr must be pre-filled with -1.  Half the scattered requests of the
// plain kernel on low-density maps -- and the number of requests is what bounds it.
// (d_m != nullptr: the row count lives on the device; `m` stays the row pitch of the table, sized for the bound)
__global__ __launch_bounds__(kBlock) void kernel_map_self_kernel(const int32_t* __restrict__ coords, int64_t m,
                                                                const int32_t* __restrict__ d_m,
                                                                const uint64_t* __restrict__ hkeys,
                                                                const int32_t* __restrict__ hvals, uint32_t mask, int step,
                                                                int32_t* __restrict__ nbr) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (d_m ? (int64_t)*d_m : m)) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[o];
    auto key_of = [&](int k, bool& ok) {
        const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
        return pack_key(c.x, c.y + dx * step, c.z + dy * step, c.w + dz * step, ok);
    };
    uint64_t got[13];
    uint32_t slot[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        bool ok;
        slot[k] = hash_key(key_of(k, ok)) & mask;
        got[k] = hkeys[slot[k]];
    }
    uint32_t hit = 0, walk = 0;
    int val[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        bool ok;
        const uint64_t key = key_of(k, ok);
        const bool h = ok && got[k] == key;
        hit |= h ? 1u << k : 0u;
        walk |= (ok && !h && got[k] != kEmptyKey) ? 1u << k : 0u;
        val[k] = hvals[slot[k] & (0u - (uint32_t)h)];
    }
#pragma unroll
    for (int k = 0; k < 13; ++k) asm volatile("" ::"v"(val[k]));
    nbr[(int64_t)13 * m + o] = (int)o;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        int res = (hit >> k) & 1u ? val[k] : -1;
        if ((walk >> k) & 1u) {
            bool ok;
            const uint64_t key = key_of(k, ok);
            uint32_t sl = (slot[k] + 1) & mask;
            for (uint32_t probe = 1; probe <= mask; ++probe) {
                const uint64_t q = hkeys[sl];
                if (q == key) { res = hvals[sl]; break; }
                if (q == kEmptyKey) break;
                sl = (sl + 1) & mask;
            }
        }
        if (res >= 0) {
            nbr[(int64_t)k * m + o] = res;
            nbr[(int64_t)(26 - k) * m + res] = (int)o;
        }
    }
}

// (d_m != nullptr: the row count lives on the device, `m` is the bound and the row pitch; rows behind the count get no pair)
__global__ void kernel_map_up_kernel(const int32_t* __restrict__ fine, const int32_t* __restrict__ parent,
                                     int64_t m, const int32_t* __restrict__ d_m, int ts, int32_t* __restrict__ nbr_up) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    if (d_m && j >= (int64_t)*d_m) {
        for (int k = 0; k < 8; ++k) nbr_up[(int64_t)k * m + j] = -1;
        return;
    }
    const int4 c = reinterpret_cast<const int4*>(fine)[j];
    const int s = 2 * ts;
    const int dx = (c.y - floor_div(c.y, s) * s) / ts;
    const int dy = (c.z - floor_div(c.z, s) * s) / ts;
    const int dz = (c.w - floor_div(c.w, s) * s) / ts;
    const int kj = dx + 2 * dy + 4 * dz;
    const int p = parent[j];
    for (int k = 0; k < 8; ++k) nbr_up[(int64_t)k * m + j] = (k == kj) ? p : -1;
}

// kernel_size-2 / stride-2 map (fine map -> its coarse map) from the parent array: fine row j is the neighbour of its parent
// under the offset given by its position inside the coarse cell.  One coalesced pass over the fine rows and one scattered
// 4-byte write each, instead of 8 table lookups per coarse row (the same table as kernel_map_kernel<2>).  nbr pre-filled -1.
__global__ void kernel_map_down_kernel(const int32_t* __restrict__ fine, const int32_t* __restrict__ parent, int64_t m_fine,
                                       const int32_t* __restrict__ d_m_fine, int ts, int64_t m_coarse,
                                       int32_t* __restrict__ nbr_down) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (d_m_fine ? min((int64_t)*d_m_fine, m_fine) : m_fine)) return;
    const int4 c = reinterpret_cast<const int4*>(fine)[j];
    const int s = 2 * ts;
    const int dx = (c.y - floor_div(c.y, s) * s) / ts;
    const int dy = (c.z - floor_div(c.z, s) * s) / ts;
    const int dz = (c.w - floor_div(c.w, s) * s) / ts;
    nbr_down[(int64_t)(dx + 2 * dy + 4 * dz) * m_coarse + parent[j]] = (int32_t)j;
}

// Morton (Z-order) key of every row at the map's own resolution: 3 x 16 interleaved bits of (x, y, z) / ts,
// the batch index above them.  Sorting rows by it gives the sparse convolution spatially compact tiles.
__device__ __forceinline__ uint64_t spread3(uint64_t v) {       // 16 bits -> every third bit
    v &= 0xffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__global__ void morton_kernel(const int32_t* __restrict__ coords, int64_t m, int ts, int64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const uint64_t x = (uint64_t)(floor_div(c.y, ts) + kKeyOff), y = (uint64_t)(floor_div(c.z, ts) + kKeyOff),
                   z = (uint64_t)(floor_div(c.w, ts) + kKeyOff);
    keys[i] = (int64_t)(((uint64_t)(c.x & 0x7fff) << 48) | spread3(x) | (spread3(y) << 1) | (spread3(z) << 2));
}

// ---------------------------------------------------------------------------------------
// ME-layout rulebook: ordered stream compaction of every offset's column of the table.
// grid = (blocks over rows, K).  counts[k*nblk + b] -> exclusive scan -> fill.
__global__ void rb_count_kernel(const int32_t* __restrict__ nbr, int64_t m_out, int32_t* __restrict__ counts) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = o < m_out && nbr[(int64_t)blockIdx.y * m_out + o] >= 0;
    const unsigned long long m = __ballot(valid);
    if (lane_id() == 0) wave_cnt[threadIdx.x / kWave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += wave_cnt[w];
        counts[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// single block: in-place exclusive scan of counts[0..n) ; counts[n] = total ;
// offset_ptr[k] = scanned counts[k*nblk], offset_ptr[K] = total.
__global__ void rb_scan_kernel(int32_t* __restrict__ counts, int n, int nblk, int k_vol,
                               int32_t* __restrict__ offset_ptr) {
    __shared__ int wave_tot[1024 / kWave];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int nw = blockDim.x / kWave;
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int v = i < n ? counts[i] : 0;
        int incl = v;
        for (int off = 1; off < kWave; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane_id() >= off) incl += t;
        }
        if (lane_id() == kWave - 1) wave_tot[threadIdx.x / kWave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < nw; ++w) {
            if (w < (int)(threadIdx.x / kWave)) wbase += wave_tot[w];
            tot += wave_tot[w];
        }
        const int carry = carry_s;
        if (i < n) counts[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[n] = carry_s;
    __syncthreads();
    for (int k = threadIdx.x; k <= k_vol; k += blockDim.x)
        offset_ptr[k] = (k == k_vol) ? counts[n] : counts[(int64_t)k * nblk];
}

__global__ void rb_fill_kernel(const int32_t* __restrict__ nbr, int64_t m_out,
                               const int32_t* __restrict__ scanned, int32_t* __restrict__ pairs_in,
                               int32_t* __restrict__ pairs_out) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = -1;
    if (o < m_out) v = nbr[(int64_t)blockIdx.y * m_out + o];
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    const int w = threadIdx.x / kWave;
    if (lane_id() == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < w; ++k) wbase += wave_cnt[k];
    if (valid) {
        const int pos = scanned[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] + wbase + popc_below(m);
        pairs_in[pos] = v;
        pairs_out[pos] = (int32_t)o;
    }
}

// ---------------------------------------------------------------------------------------
// Tail map of a kernel_size-3 / stride-1 kernel map (include/lidiff_amd.h, lidiff_tail_map): the pairs of all offsets
// but `skip` (the centre = identity) as P output rows sorted by (offset, output row) -- tail_nbr[k][p] = input row of pair p
// at its own offset, -1 elsewhere -- plus a CSR over the map's output rows listing each row's pairs in ascending offset.
// (d_m != nullptr: the row count lives on the device and m_out is the row pitch of the table / the bound of the grid)
__global__ void tail_count_kernel(const int32_t* __restrict__ nbr, int64_t m_out, const int32_t* __restrict__ d_m, int skip,
                                  int32_t* __restrict__ counts) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = d_m ? (int64_t)*d_m : m_out;
    const bool valid = (int)blockIdx.y != skip && o < rows && nbr[(int64_t)blockIdx.y * m_out + o] >= 0;
    const unsigned long long m = __ballot(valid);
    if (lane_id() == 0) wave_cnt[threadIdx.x / kWave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += wave_cnt[w];
        counts[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// per-row pair counts of the tail map: FINAL == false: per-block sums only; FINAL == true: row_ptr[o] = blk_base[block] +
// exclusive scan inside the block, row_ptr[m_out] = total (the block sums are scanned in between by scan_i32_kernel)
template <bool FINAL>
__global__ void tail_rowcnt_kernel(const int32_t* __restrict__ nbr, int64_t m_out, const int32_t* __restrict__ d_m, int k_vol,
                                   int skip, int32_t* __restrict__ blk_sums, int32_t* __restrict__ row_ptr) {
    __shared__ int wave_tot[kWavesPerBlock];
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = d_m ? (int64_t)*d_m : m_out;
    int c = 0;
    if (o < rows)
        for (int k = 0; k < k_vol; ++k) c += (k != skip && nbr[(int64_t)k * m_out + o] >= 0) ? 1 : 0;
    int incl = c;
    for (int off = 1; off < kWave; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane_id() >= off) incl += t;
    }
    const int w = threadIdx.x / kWave;
    if (lane_id() == kWave - 1) wave_tot[w] = incl;
    __syncthreads();
    int wbase = 0, tot = 0;
    for (int q = 0; q < kWavesPerBlock; ++q) {
        if (q < w) wbase += wave_tot[q];
        tot += wave_tot[q];
    }
    if constexpr (!FINAL) {
        if (threadIdx.x == 0) blk_sums[blockIdx.x] = tot;
    } else {
        const int base = blk_sums[blockIdx.x];
        if (o < m_out) row_ptr[o] = base + wbase + incl - c;
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) row_ptr[m_out] = base + tot;
    }
}

// single block: in-place exclusive scan of data[0..n), data[n] = total
__global__ void scan_i32_kernel(int32_t* __restrict__ data, int64_t n) {
    __shared__ int wave_tot[1024 / kWave];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int nw = blockDim.x / kWave;
    for (int64_t base = 0; base < n; base += blockDim.x) {
        const int64_t i = base + threadIdx.x;
        const int v = i < n ? data[i] : 0;
        int incl = v;
        for (int off = 1; off < kWave; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane_id() >= off) incl += t;
        }
        if (lane_id() == kWave - 1) wave_tot[threadIdx.x / kWave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < nw; ++w) {
            if (w < (int)(threadIdx.x / kWave)) wbase += wave_tot[w];
            tot += wave_tot[w];
        }
        const int carry = carry_s;
        if (i < n) data[i] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) data[n] = carry_s;
}

__global__ void tail_fill_kernel(const int32_t* __restrict__ nbr, int64_t m_out, const int32_t* __restrict__ d_m, int skip,
                                 int64_t n_pairs, const int32_t* __restrict__ scanned, const int32_t* __restrict__ row_ptr,
                                 int32_t* __restrict__ tail_nbr, int32_t* __restrict__ idx) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int k = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = d_m ? (int64_t)*d_m : m_out;
    int v = -1;
    if (k != skip && o < rows) v = nbr[(int64_t)k * m_out + o];
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    const int w = threadIdx.x / kWave;
    if (lane_id() == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < w; ++q) wbase += wave_cnt[q];
    if (valid) {
        const int pos = scanned[(int64_t)k * gridDim.x + blockIdx.x] + wbase + popc_below(m);
        tail_nbr[(int64_t)k * n_pairs + pos] = v;
        int rank = 0;                                            // pairs of this output row at lower offsets
        for (int q = 0; q < k; ++q) rank += (q != skip && nbr[(int64_t)q * m_out + o] >= 0) ? 1 : 0;
        idx[row_ptr[o] + rank] = pos;
    }
}

// The same fill into a pair list whose length is only BOUNDED on the host (host-read-free steps: the pair count stays on the
// device): pair_in[pos] = the pair's input row (what lidiff_spconv_fwd_pairs walks -- the [K, P] table form is not made),
// idx as above.  Pairs behind the bound are dropped and LIDIFF_STATUS_BOUND raised in *d_status: the caller redoes the step
// with exact sizes; consumers clamp their ranges to the bound, so nothing is overrun meanwhile.  idx must be pre-zeroed.
__global__ void tail_fill_bounded_kernel(const int32_t* __restrict__ nbr, int64_t m_out, const int32_t* __restrict__ d_m, int skip,
                                         int k_vol, int64_t n_bound, const int32_t* __restrict__ scanned,
                                         const int32_t* __restrict__ offset_ptr, const int32_t* __restrict__ row_ptr,
                                         int32_t* __restrict__ pair_in, int32_t* __restrict__ idx, int32_t* __restrict__ d_status) {
    __shared__ int wave_cnt[kWavesPerBlock];
    const int k = blockIdx.y;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows = d_m ? min((int64_t)*d_m, m_out) : m_out;
    if (blockIdx.x == 0 && k == 0 && threadIdx.x == 0 && (int64_t)offset_ptr[k_vol] > n_bound) atomicOr(d_status, LIDIFF_STATUS_BOUND);
    int v = -1;
    if (k != skip && o < rows) v = nbr[(int64_t)k * m_out + o];
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    const int w = threadIdx.x / kWave;
    if (lane_id() == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < w; ++q) wbase += wave_cnt[q];
    if (valid) {
        const int64_t pos = (int64_t)scanned[(int64_t)k * gridDim.x + blockIdx.x] + wbase + popc_below(m);
        if (pos < n_bound) {
            pair_in[pos] = v;
            int rank = 0;                                            // pairs of this output row at lower offsets
            for (int q = 0; q < k; ++q) rank += (q != skip && nbr[(int64_t)q * m_out + o] >= 0) ? 1 : 0;
            const int64_t slot = (int64_t)row_ptr[o] + rank;
            if (slot < n_bound) idx[slot] = (int32_t)pos;
        }
    }
}

// counts / status words of a coordinate pyramid -> host-visible (pinned, device-mapped) memory, then a sequence number: the
// host learns the sizes of step i while step i + 1 is already queued, without a copy or a synchronisation on its side
__global__ void publish_kernel(const int32_t* __restrict__ words, int n, const int32_t* __restrict__ d_status,
                               volatile int32_t* __restrict__ host, int32_t seq) {
    const int i = threadIdx.x;
    if (i < n) host[2 + i] = words[i];
    if (i == 0) host[1] = d_status ? *d_status : 0;
    __threadfence_system();
    __syncthreads();
    if (i == 0) {
        host[0] = seq;
        __threadfence_system();
    }
}

// ---------------------------------------------------------------------------------------
// row gather / scatter-add
template <bool VEC4>
__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                   int64_t n, int c, float* __restrict__ dst) {
    const int cw = VEC4 ? c / 4 : c;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * cw) return;
    const int64_t r = e / cw;
    const int j = (int)(e % cw);
    const int64_t s = idx[r];
    if (VEC4)
        reinterpret_cast<float4*>(dst)[r * cw + j] = reinterpret_cast<const float4*>(src)[s * cw + j];
    else
        dst[r * c + j] = src[s * c + j];
}

// out[r, :] = leaky_relu(src[idx[r], :] + bias[:], slope): the conditioning path's hidden layer
// (minkunet.py:424-431 after commuting the row-wise MLP with the gather): one pass instead of gather + add + activation.
// dst[r,:] = x[r,:] * table[idx[r],:] -- the conditioning multiply with the whole MLP commuted in front of the gather
// (idx == nullptr: every row takes table row 0 -- the one-voxel unconditional branch; d_n: the valid rows live on the device)
__global__ void gather_mul_rows_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                       const int64_t* __restrict__ idx, int64_t n, const int32_t* __restrict__ d_n, int c4,
                                       float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (d_n ? min((int64_t)*d_n, n) : n) * c4) return;
    const int64_t r = e / c4;
    const int j = (int)(e % c4);
    const float4 v = reinterpret_cast<const float4*>(x)[e];
    const float4 w = reinterpret_cast<const float4*>(table)[(idx ? idx[r] : 0) * c4 + j];
    reinterpret_cast<float4*>(dst)[e] = make_float4(v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
}

__global__ void gather_bias_leaky_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                         const float* __restrict__ bias, int64_t n, int c4, float slope,
                                         float* __restrict__ dst) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * c4) return;
    const int64_t r = e / c4;
    const int j = (int)(e % c4);
    const float4 v = reinterpret_cast<const float4*>(src)[idx[r] * c4 + j];
    const float4 b = reinterpret_cast<const float4*>(bias)[j];
    float4 o;
    o.x = v.x + b.x; o.y = v.y + b.y; o.z = v.z + b.z; o.w = v.w + b.w;
    o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
    o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
    reinterpret_cast<float4*>(dst)[r * c4 + j] = o;
}


// dst[o] = sum of src[order[q]] for q in [ptr[o], ptr[o + 1]) -- the scatter-add over a destination-sorted source list: every
// destination row adds its sources in list order (the stable sort keeps source order), no atomics: deterministic.
// One thread per (destination, float4) walks its segment.  A segment longer than kSegShort sources is NOT summed that way (the
// unconditional branch of a training step -- models.py:192-195, a part of all zeros -- puts ~180 000 sources on each of 2
// destinations at every conditioning level: 256 threads would walk them one by one): its first thread registers the destination
// and reserves ceil(len / kSegChunk) chunks; segment_sum_chunk_kernel sums every chunk of kSegChunk sources with one workgroup per
// (chunk, 32 channels) -- lane r adds sources r, r + 32, ... in order, the 32 lane sums are added in lane order -- and
// segment_sum_combine_kernel adds a destination's chunk sums in chunk order.  The order in which destinations register decides
// only WHERE partial sums are stored, never the order of a sum: deterministic.
constexpr int kSegShort = 64;
constexpr int kSegChunk = 1024;
struct SegWork {                     // workspace layout (int32 words): [0] long segments, [1] chunks, then the two lists
    int32_t* head; int32_t* longs; int32_t* chunks; float* part; int64_t cap_long, cap_chunks;
};
__host__ __device__ inline int64_t seg_cap_long(int64_t n) { return n / kSegShort + 1; }
__host__ __device__ inline int64_t seg_cap_chunks(int64_t n) { return n / kSegChunk + seg_cap_long(n); }
static SegWork seg_work(void* ws, int64_t n, int c) {
    SegWork w;
    w.cap_long = seg_cap_long(n); w.cap_chunks = seg_cap_chunks(n);
    w.head = (int32_t*)ws;
    w.longs = w.head + 4;                                   // (destination, first chunk, chunks) per long segment
    w.chunks = w.longs + 3 * w.cap_long;                    // (long-segment slot, chunk index) per chunk
    w.part = (float*)(((uintptr_t)(w.chunks + 2 * w.cap_chunks) + 15) & ~(uintptr_t)15);
    return w;
}

template <bool VEC4>
__global__ void segment_sum_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ order,
                                        const int64_t* __restrict__ ptr, int64_t m, int c, float* __restrict__ dst, SegWork w) {
    const int cw = VEC4 ? c / 4 : c;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * cw) return;
    const int64_t o = e / cw;
    const int j = (int)(e % cw);
    const int64_t lo = ptr[o], hi = ptr[o + 1];
    if (w.head != nullptr && hi - lo > kSegShort) {
        if (j == 0) {
            const int nch = (int)((hi - lo + kSegChunk - 1) / kSegChunk);
            const int slot = atomicAdd(&w.head[0], 1);
            const int base = atomicAdd(&w.head[1], nch);
            w.longs[3 * slot] = (int32_t)o; w.longs[3 * slot + 1] = base; w.longs[3 * slot + 2] = nch;
            for (int i = 0; i < nch; ++i) { w.chunks[2 * (base + i)] = slot; w.chunks[2 * (base + i) + 1] = i; }
        }
        return;
    }
    if constexpr (VEC4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t q = lo; q < hi; ++q) {
            const float4 v = reinterpret_cast<const float4*>(src + order[q] * c)[j];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4*>(dst + o * c)[j] = a;
    } else {
        float a = 0.f;
        for (int64_t q = lo; q < hi; ++q) a += src[order[q] * c + j];
        dst[o * c + j] = a;
    }
}

// workgroup (x, g): chunks x, x + gridDim.x, ... of the registered long segments, channels [32 g, 32 g + 32); 256 threads =
// 32 source lanes x 8 channel quads (128 contiguous bytes per source row)
__global__ __launch_bounds__(256) void segment_sum_chunk_kernel(const float* __restrict__ src, const int64_t* __restrict__ order,
                                                                const int64_t* __restrict__ ptr, int c, SegWork w) {
    __shared__ float sm[32][33];
    const int nchunks = w.head[1];
    const int r = threadIdx.x >> 3, q = threadIdx.x & 7;
    const int ch0 = 32 * blockIdx.y + 4 * q;
    for (int i = blockIdx.x; i < nchunks; i += gridDim.x) {
        const int slot = w.chunks[2 * i], ci = w.chunks[2 * i + 1];
        const int64_t o = w.longs[3 * slot];
        const int64_t lo = ptr[o] + (int64_t)ci * kSegChunk, hi = min(ptr[o + 1], lo + kSegChunk);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t s = lo + r; s < hi; s += 32) {
            const float* row = src + order[s] * c;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ch0 + e < c) a[e] += row[ch0 + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sm[r][4 * q + e] = a[e];
        __syncthreads();
        if (threadIdx.x < 32 && 32 * blockIdx.y + threadIdx.x < c) {
            float t = 0.f;
            for (int rr = 0; rr < 32; ++rr) t += sm[rr][threadIdx.x];
            w.part[(int64_t)i * c + 32 * blockIdx.y + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// one thread per (long segment, channel): its chunk sums in chunk order
__global__ void segment_sum_combine_kernel(int c, float* __restrict__ dst, SegWork w) {
    const int nlong = w.head[0];
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)nlong * c; e += (int64_t)gridDim.x * blockDim.x) {
        const int slot = (int)(e / c), ch = (int)(e % c);
        const int base = w.longs[3 * slot + 1], nch = w.longs[3 * slot + 2];
        float t = 0.f;
        for (int i = 0; i < nch; ++i) t += w.part[(int64_t)(base + i) * c + ch];
        dst[(int64_t)w.longs[3 * slot] * c + ch] = t;
    }
}

// ---------------------------------------------------------------------------------------
// nearest part voxel: one full row per lane, part rows streamed through LDS tiles; fp32
// arithmetic as pykeops does (exact on integer coordinates below 2^12), strict '<' while
// scanning j ascending => lowest index wins ties.
constexpr int kMatchTile = 1024;
constexpr int kMatchPt = 2;                               // full rows per lane: every LDS read feeds two evaluations
// blockIdx.y splits the part rows when the full rows alone cannot fill the chip (SPLIT): the partial winners meet
// in idx[] through a 64-bit atomic min on (distance bits << 32 | row) -- non-negative floats order like their
// bits, equal distances fall to the lower row -- and nn_match_finish_kernel strips the distance.
// F32IN: the rows are float4 already (the generic pykeops-style arg-min of lidiff_argmin_rows_f32: no batch scale).
template <bool SPLIT, bool F32IN = false>
__global__ void nn_match_kernel(const int32_t* __restrict__ full, int64_t m_full,
                                const int32_t* __restrict__ part, int64_t m_part, int64_t part_per_split,
                                const int32_t* __restrict__ d_max_coord, int64_t* __restrict__ idx,
                                const int32_t* __restrict__ d_m_full = nullptr, const int32_t* __restrict__ d_gate = nullptr,
                                int by_batch = 0) {
    __shared__ float4 tile[kMatchTile];
    __shared__ int seg[4];
    if (d_gate != nullptr && *d_gate == 0) return;        // the unrestricted pass behind a by-batch one: only when that asked for it
    if (d_m_full != nullptr) {                            // row count still on the device (a pyramid before its host read):
        m_full = *d_m_full;                               // the grid covers the bound, workgroups beyond the count leave at once
        if ((int64_t)blockIdx.x * blockDim.x * kMatchPt >= m_full) return;
    }
    const float scale = F32IN ? 1.0f : 2.0f * (float)(*d_max_coord);
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x * kMatchPt + threadIdx.x;   // rows i0, i0 + blockDim.x
    static_assert(kMatchPt == 2, "the two rows of a lane ride in the halves of packed-fp32 registers");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 fb, fx, fy, fz;                                 // component q = row q of this lane (v_pk_add / v_pk_fma)
    float best[kMatchPt];
    int best_j[kMatchPt];
#pragma unroll
    for (int q = 0; q < kMatchPt; ++q) {
        const int64_t i = min(i0 + (int64_t)q * blockDim.x, m_full - 1);
        if constexpr (F32IN) {
            const float4 c = reinterpret_cast<const float4*>(full)[i];
            fb[q] = c.x; fx[q] = c.y; fy[q] = c.z; fz[q] = c.w;
        } else {
            const int4 c = reinterpret_cast<const int4*>(full)[i];
            fb[q] = (float)c.x * scale; fx[q] = (float)c.y; fy[q] = (float)c.z; fz[q] = (float)c.w;
        }
        best[q] = INFINITY; best_j[q] = 0;
    }
    int64_t lo = SPLIT ? (int64_t)blockIdx.y * part_per_split : 0;
    int64_t hi = SPLIT ? min(m_part, lo + part_per_split) : m_part;
    if (by_batch && !F32IN) {
        // by-batch pass: only the part rows of the batch elements this workgroup's rows belong to (part rows grouped by ascending
        // batch index: lower / upper bound of the batch column).  Conclusive for a row whose best distance stays below scale^2 --
        // the batch term alone of any other element's row; lidiff_nn_match checks that (and the grouping) and runs the
        // unrestricted pass over the same idx[] when it is not.
        if (threadIdx.x == 0) { seg[0] = INT32_MAX; seg[1] = INT32_MIN; }
        __syncthreads();
        int bmin = INT32_MAX, bmax = INT32_MIN;
#pragma unroll
        for (int q = 0; q < kMatchPt; ++q) {
            const int b = reinterpret_cast<const int4*>(full)[min(i0 + (int64_t)q * blockDim.x, m_full - 1)].x;
            bmin = min(bmin, b); bmax = max(bmax, b);
        }
        for (int off = kWave / 2; off > 0; off >>= 1) { bmin = min(bmin, __shfl_down(bmin, off)); bmax = max(bmax, __shfl_down(bmax, off)); }
        if (lane_id() == 0) { atomicMin(&seg[0], bmin); atomicMax(&seg[1], bmax); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int b0 = seg[0], b1 = seg[1];
            int64_t a = 0, z = m_part;                    // first row with batch >= b0
            while (a < z) { const int64_t mid = (a + z) >> 1; if (reinterpret_cast<const int4*>(part)[mid].x < b0) a = mid + 1; else z = mid; }
            seg[2] = (int)a;
            z = m_part;                                   // first row with batch > b1
            while (a < z) { const int64_t mid = (a + z) >> 1; if (reinterpret_cast<const int4*>(part)[mid].x <= b1) a = mid + 1; else z = mid; }
            seg[3] = (int)a;
        }
        __syncthreads();
        lo = max(lo, (int64_t)seg[2]);
        hi = min(hi, (int64_t)seg[3]);
    }
    for (int64_t base = lo; base < hi; base += kMatchTile) {
        const int cnt = (int)min((int64_t)kMatchTile, hi - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            if constexpr (F32IN) {
                tile[t] = reinterpret_cast<const float4*>(part)[base + t];
            } else {
                const int4 c = reinterpret_cast<const int4*>(part)[base + t];
                tile[t] = make_float4((float)c.x * scale, (float)c.y, (float)c.z, (float)c.w);
            }
        }
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const float4 p = tile[t];
            const f32x2 db = fb - p.x, dx = fx - p.y, dy = fy - p.z, dz = fz - p.w;
            const f32x2 d = db * db + dx * dx + dy * dy + dz * dz;       // same operation order as the scalar form
#pragma unroll
            for (int q = 0; q < kMatchPt; ++q)
                if (d[q] < best[q]) { best[q] = d[q]; best_j[q] = (int)(base + t); }
        }
    }
#pragma unroll
    for (int q = 0; q < kMatchPt; ++q) {
        const int64_t i = i0 + (int64_t)q * blockDim.x;
        if (i >= m_full) continue;
        if constexpr (SPLIT) {
            if (lo < hi)
                atomicMin(reinterpret_cast<unsigned long long*>(idx) + i,
                          ((unsigned long long)__float_as_uint(best[q]) << 32) | (unsigned int)best_j[q]);
        } else {
            idx[i] = best_j[q];
        }
    }
}

// behind a by-batch pass: is every row's winner conclusive (distance below scale^2, the batch term of any other element's rows),
// and are the part rows grouped by ascending batch index at all?  *d_gate = 1 otherwise: the unrestricted pass runs.
__global__ void nn_match_check_kernel(const int64_t* __restrict__ idx, int64_t m_full, const int32_t* __restrict__ part, int64_t m_part,
                                      const int32_t* __restrict__ d_max_coord, int32_t* __restrict__ d_gate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float scale = 2.0f * (float)(*d_max_coord);
    bool bad = false;
    if (i < m_full) {
        const unsigned long long v = (unsigned long long)idx[i];
        bad = v == ~0ull || !(__uint_as_float((unsigned)(v >> 32)) < scale * scale);
    }
    if (i + 1 < m_part) bad |= reinterpret_cast<const int4*>(part)[i].x > reinterpret_cast<const int4*>(part)[i + 1].x;
    if (__ballot(bad) != 0ull && lane_id() == 0) atomicOr(d_gate, 1);
}

__global__ void nn_match_finish_kernel(int64_t* __restrict__ idx, int64_t m_full, const int32_t* __restrict__ d_m_full = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_m_full != nullptr) m_full = *d_m_full;
    if (i < m_full) idx[i] &= 0xffffffffll;
}

// max over ALL columns of the first *d_m rows (torch's full.max() of minkunet.py:410, with the row count on the device);
// *d_max starts at a value below every coordinate
__global__ void coord_max_dev_kernel(const int32_t* __restrict__ coords, const int32_t* __restrict__ d_m, int32_t* __restrict__ d_max) {
    const int64_t m = *d_m;
    int best = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
        const int4 c = reinterpret_cast<const int4*>(coords)[i];
        best = max(best, max(max(c.x, c.y), max(c.z, c.w)));
    }
    for (int off = kWave / 2; off > 0; off >>= 1) best = max(best, __shfl_down(best, off));
    __shared__ int wbest[kBlock / kWave];                 // one atomic per workgroup, not per wave
    if (lane_id() == 0) wbest[threadIdx.x / kWave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x / kWave); ++w) best = max(best, wbest[w]);
        if (best != INT32_MIN) atomicMax(d_max, best);
    }
}

// ---------------------------------------------------------------------------------------
// Farthest-point sampling of the input scan -- DiffCompletion.preprocess_scan (pipeline:92-105; open3d
// farthest_point_down_sample): start from point 0; every step lowers each point's distance to the selected set by
// the newest selection and picks the farthest point (first maximum).  float64 like the reference.  One launch per
// selection, no host round trip: every block reduces its points, the last block to finish (device-wide counter)
// reduces the per-block maxima and publishes the next selection.
constexpr int kFpsBlock = 1024;
__global__ __launch_bounds__(kFpsBlock) void fps_step_kernel(const double* __restrict__ pts, double* __restrict__ dist,
                                                            int64_t n, int64_t* __restrict__ sel, int64_t step,
                                                            double* __restrict__ blk_val, int64_t* __restrict__ blk_idx,
                                                            unsigned int* __restrict__ counter) {
    __shared__ double s_val[kFpsBlock / kWave];
    __shared__ int64_t s_idx[kFpsBlock / kWave];
    __shared__ bool is_last;
    const int64_t cur = sel[step];                         // newest selection
    const double sx = pts[3 * cur], sy = pts[3 * cur + 1], sz = pts[3 * cur + 2];
    const int64_t i = (int64_t)blockIdx.x * kFpsBlock + threadIdx.x;
    double best = -1.0;
    int64_t best_i = 0x7fffffffffffffffll;
    if (i < n) {
        const double dx = pts[3 * i] - sx, dy = pts[3 * i + 1] - sy, dz = pts[3 * i + 2] - sz;
        const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));   // no fma contraction
        const double nd = fmin(dist[i], d);
        dist[i] = nd;
        best = nd;
        best_i = i;
    }
    auto better = [](double v, int64_t j, double bv, int64_t bj) { return v > bv || (v == bv && j < bj); };
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off);
        const int64_t oj = __shfl_down(best_i, off);
        if (better(ov, oj, best, best_i)) { best = ov; best_i = oj; }
    }
    if (lane_id() == 0) { s_val[threadIdx.x / kWave] = best; s_idx[threadIdx.x / kWave] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kFpsBlock / kWave; ++w)
            if (better(s_val[w], s_idx[w], best, best_i)) { best = s_val[w]; best_i = s_idx[w]; }
        blk_val[blockIdx.x] = best;
        blk_idx[blockIdx.x] = best_i;
        __threadfence();                                   // publish before taking the ticket
        is_last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();                                       // acquire the other blocks' maxima
    best = -1.0;
    best_i = 0x7fffffffffffffffll;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += kFpsBlock) {
        const double v = __hip_atomic_load(&blk_val[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t j = __hip_atomic_load(&blk_idx[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (better(v, j, best, best_i)) { best = v; best_i = j; }
    }
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const double ov = __shfl_down(best, off);
        const int64_t oj = __shfl_down(best_i, off);
        if (better(ov, oj, best, best_i)) { best = ov; best_i = oj; }
    }
    if (lane_id() == 0) { s_val[threadIdx.x / kWave] = best; s_idx[threadIdx.x / kWave] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kFpsBlock / kWave; ++w)
            if (better(s_val[w], s_idx[w], best, best_i)) { best = s_val[w]; best_i = s_idx[w]; }
        sel[step + 1] = best_i;
        *counter = 0;
    }
}

// The same sampling as ONE persistent cooperative launch: every workgroup keeps the running distances of its points in
// registers for the whole run; per selection it publishes its maximum, all workgroups meet at a device-wide barrier (a
// monotonically increasing ticket counter, agent-scope release / acquire), and every workgroup reduces the published
// maxima itself -- all arrive at the same winner, nobody waits for a broadcast.  Partial maxima are double-buffered by the
// parity of the selection.  17 999 launches of ~11 us become 17 999 barriers of ~7 us (measured: 134 ms against 198 ms).  The spin is bounded: a
// workgroup that waits longer than kFpsSpinLimit polls sets *status and leaves (the host then falls back to the
// one-launch-per-selection kernel).
constexpr int kFpsCoopPt = 8;                 // points per thread (registers); n <= grid x 1024 x 8
constexpr unsigned kFpsSpinLimit = 1u << 22;
__global__ __launch_bounds__(kFpsBlock) void fps_coop_kernel(const double* __restrict__ pts, int64_t n, int64_t n_samples,
                                                            int pt, int64_t* __restrict__ sel, double* part_val,
                                                            int64_t* part_idx, unsigned int* counter, int* status) {
    __shared__ double s_val[kFpsBlock / kWave];
    __shared__ int64_t s_idx[kFpsBlock / kWave];
    __shared__ int64_t s_cur;
    __shared__ int s_abort;
    const int G = (int)gridDim.x;
    const int64_t gtid = (int64_t)blockIdx.x * kFpsBlock + threadIdx.x, stride = (int64_t)G * kFpsBlock;
    double px[kFpsCoopPt], py[kFpsCoopPt], pz[kFpsCoopPt], dmin[kFpsCoopPt];
#pragma unroll
    for (int q = 0; q < kFpsCoopPt; ++q) {
        const int64_t i = gtid + q * stride;
        const bool ok = q < pt && i < n;
        px[q] = ok ? pts[3 * i] : 0.0;
        py[q] = ok ? pts[3 * i + 1] : 0.0;
        pz[q] = ok ? pts[3 * i + 2] : 0.0;
        dmin[q] = __longlong_as_double(0x7f7f7f7f7f7f7f7fll);          // as lidiff_fps' memset: 1.4e306
    }
    if (threadIdx.x == 0) s_abort = 0;
    if (gtid == 0) sel[0] = 0;
    auto better = [](double v, int64_t j, double bv, int64_t bj) { return v > bv || (v == bv && j < bj); };
    int64_t cur = 0;
    for (int64_t step = 0; step + 1 < n_samples; ++step) {
        const double sx = pts[3 * cur], sy = pts[3 * cur + 1], sz = pts[3 * cur + 2];
        double best = -1.0;
        int64_t best_i = 0x7fffffffffffffffll;
#pragma unroll
        for (int q = 0; q < kFpsCoopPt; ++q) {
            const int64_t i = gtid + q * stride;
            if (q < pt && i < n) {
                const double dx = px[q] - sx, dy = py[q] - sy, dz = pz[q] - sz;
                const double d = __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
                const double nd = fmin(dmin[q], d);
                dmin[q] = nd;
                if (better(nd, i, best, best_i)) { best = nd; best_i = i; }
            }
        }
        for (int off = kWave / 2; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off);
            const int64_t oj = __shfl_down(best_i, off);
            if (better(ov, oj, best, best_i)) { best = ov; best_i = oj; }
        }
        if (lane_id() == 0) { s_val[threadIdx.x / kWave] = best; s_idx[threadIdx.x / kWave] = best_i; }
        __syncthreads();
        const int par = (int)(step & 1);
        if (threadIdx.x == 0) {
            for (int w = 1; w < kFpsBlock / kWave; ++w)
                if (better(s_val[w], s_idx[w], best, best_i)) { best = s_val[w]; best_i = s_idx[w]; }
            __hip_atomic_store(&part_val[par * G + blockIdx.x], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&part_idx[par * G + blockIdx.x], best_i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(step + 1) * (unsigned)G;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kFpsSpinLimit ||
                    __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (s_abort) return;
        if (threadIdx.x < kWave) {                              // every workgroup reduces all published maxima
            best = -1.0;
            best_i = 0x7fffffffffffffffll;
            for (int b = threadIdx.x; b < G; b += kWave) {
                const double v = __hip_atomic_load(&part_val[par * G + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int64_t j = __hip_atomic_load(&part_idx[par * G + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (better(v, j, best, best_i)) { best = v; best_i = j; }
            }
            for (int off = kWave / 2; off > 0; off >>= 1) {
                const double ov = __shfl_down(best, off);
                const int64_t oj = __shfl_down(best_i, off);
                if (better(ov, oj, best, best_i)) { best = ov; best_i = oj; }
            }
            if (threadIdx.x == 0) {
                s_cur = best_i;
                if (blockIdx.x == 0) sel[step + 1] = best_i;
            }
        }
        __syncthreads();
        cur = s_cur;
    }
}

// ---------------------------------------------------------------------------------------
// nearest neighbour of every point of a in b (3-D, squared Euclidean distance, T = float or double): the distance
// queries behind the evaluation metrics and the Chamfer loss.  Exhaustive and exact: kNnPt query points per lane,
// b streamed through LDS tiles (broadcast reads), b split over blockIdx.y when a alone cannot fill the chip; the
// partial winners of the splits are merged in ascending split order with strict '<', so the lowest b index wins ties.
constexpr int kNnTile = 1024;
constexpr int kNnPt = 2;
// the squared distance as BOTH nearest-neighbour kernels form it (one expression, one contraction pattern: equal bits)
template <typename T>
__device__ __forceinline__ T nn_d2(T ax, T ay, T az, T bx, T by, T bz) {
    const T dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}

template <typename T>
__global__ void nn_dist_kernel(const T* __restrict__ a, int64_t n, const T* __restrict__ b, int64_t m,
                               int64_t m_per_split, T* __restrict__ part_d2, int32_t* __restrict__ part_idx) {
    __shared__ T tx[kNnTile], ty[kNnTile], tz[kNnTile];
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * kNnPt;
    T ax[kNnPt], ay[kNnPt], az[kNnPt], best[kNnPt];
    int32_t best_j[kNnPt];
#pragma unroll
    for (int q = 0; q < kNnPt; ++q) {
        const int64_t i = min(i0 + q, n - 1);
        ax[q] = a[3 * i]; ay[q] = a[3 * i + 1]; az[q] = a[3 * i + 2];
        best[q] = (T)INFINITY; best_j[q] = 0;
    }
    const int64_t lo = (int64_t)blockIdx.y * m_per_split, hi = min(m, lo + m_per_split);
    for (int64_t base = lo; base < hi; base += kNnTile) {
        const int cnt = (int)min((int64_t)kNnTile, hi - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            tx[t] = b[3 * (base + t)]; ty[t] = b[3 * (base + t) + 1]; tz[t] = b[3 * (base + t) + 2];
        }
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            const T bx = tx[t], by = ty[t], bz = tz[t];
#pragma unroll
            for (int q = 0; q < kNnPt; ++q) {
                const T d = nn_d2(ax[q], ay[q], az[q], bx, by, bz);
                if (d < best[q]) { best[q] = d; best_j[q] = (int32_t)(base + t); }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kNnPt; ++q)
        if (i0 + q < n) {
            part_d2[(int64_t)blockIdx.y * n + i0 + q] = best[q];
            part_idx[(int64_t)blockIdx.y * n + i0 + q] = best_j[q];
        }
}

template <typename T>
__global__ void nn_dist_merge_kernel(const T* __restrict__ part_d2, const int32_t* __restrict__ part_idx, int64_t n,
                                     int splits, T* __restrict__ d2, int64_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T best = part_d2[i];
    int32_t j = part_idx[i];
    for (int s = 1; s < splits; ++s) {
        const T d = part_d2[(int64_t)s * n + i];
        if (d < best) { best = d; j = part_idx[(int64_t)s * n + i]; }
    }
    d2[i] = best;
    idx[i] = j;
}

// ---------------------------------------------------------------------------------------
// The same nearest neighbour through a uniform grid over the searched cloud (the refinement network's Chamfer
// loss, models_refine.py:72, compares 6 x 180 000 predicted with 2 x 180 000 target points per batch item: 3.9e11 distance
// evaluations per direction for the exhaustive kernel).  b's points are binned into cubic cells of edge `cell` -- the cells are
// the voxel hash of coords.hip (insert / flag / scan, 16-bit cell coordinates) -- and stored cell by cell; a query walks the
// cubic shells of cells around its own cell, nearest first, and stops once its best distance is below the nearest possible
// point of every unvisited shell.  EXACT: the same squared distances (nn_d2), the lowest index on ties (a tie in a farther
// shell is still visited: the stop test is strict and takes 0.5 % off the shell's distance for the rounding of the cell
// assignment) -- bit-identical (d2, idx) to nn_dist_kernel.  Queries that have not finished after kGridShells shells (far from
// every point of b) and clouds whose cells leave the 16-bit key range go through the exhaustive kernel.
constexpr int kGridShells = 6;

template <typename T>
__device__ __forceinline__ int grid_cell(T x, T inv_cell) { return (int)floor((double)x * (double)inv_cell); }

template <typename T>
__global__ void grid_cells_kernel(const T* __restrict__ b, int64_t m, T inv_cell, int32_t* __restrict__ cells) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    reinterpret_cast<int4*>(cells)[j] = make_int4(0, grid_cell(b[3 * j], inv_cell), grid_cell(b[3 * j + 1], inv_cell),
                                                  grid_cell(b[3 * j + 2], inv_cell));
}

__global__ void grid_count_kernel(const int32_t* __restrict__ cell_of, int64_t m, int32_t* __restrict__ cnt) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) atomicAdd(&cnt[cell_of[j]], 1);
}

// points into their cell's slice (the order inside a cell is whatever the atomics give: the search breaks ties by index)
template <typename T>
__global__ void grid_fill_kernel(const T* __restrict__ b, const int32_t* __restrict__ cell_of, int64_t m,
                                 const int32_t* __restrict__ start, int32_t* __restrict__ cursor, T* __restrict__ sx,
                                 int32_t* __restrict__ sj) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int c = cell_of[j];
    const int pos = start[c] + atomicAdd(&cursor[c], 1);
    sx[3 * (int64_t)pos] = b[3 * j]; sx[3 * (int64_t)pos + 1] = b[3 * j + 1]; sx[3 * (int64_t)pos + 2] = b[3 * j + 2];
    sj[pos] = (int32_t)j;
}

template <typename T>
__global__ void grid_query_kernel(const T* __restrict__ a, int64_t n, T cell, T inv_cell, const uint64_t* __restrict__ hkeys,
                                  const int32_t* __restrict__ hvals, uint32_t mask, const int32_t* __restrict__ start,
                                  const T* __restrict__ sx, const int32_t* __restrict__ sj, const int32_t* __restrict__ d_status,
                                  T* __restrict__ d2, int64_t* __restrict__ idx, int32_t* __restrict__ left, int32_t* __restrict__ n_left) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T ax = a[3 * i], ay = a[3 * i + 1], az = a[3 * i + 2];
    T best = (T)INFINITY;
    int32_t best_j = 0x7fffffff;
    bool done = false;
    // (a cloud whose cells left the key range was not binned completely: everything goes to the exhaustive pass)
    if ((*d_status & LIDIFF_STATUS_KEY_RANGE) == 0 && ax == ax && ay == ay && az == az) {
        const int cx = grid_cell(ax, inv_cell), cy = grid_cell(ay, inv_cell), cz = grid_cell(az, inv_cell);
        for (int r = 0; r <= kGridShells && !done; ++r) {
            for (int dz = -r; dz <= r; ++dz)
                for (int dy = -r; dy <= r; ++dy) {
                    const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                    for (int dx = -r; dx <= r; dx += (face || r == 0) ? 1 : 2 * r) {      // interior rows: the two end cells only
                        bool ok;
                        const uint64_t key = pack_key(0, cx + dx, cy + dy, cz + dz, ok);
                        if (!ok) continue;
                        const int c = hash_find(hkeys, hvals, mask, key);
                        if (c < 0) continue;
                        for (int q = start[c], qe = start[c + 1]; q < qe; ++q) {
                            const T d = nn_d2(ax, ay, az, sx[3 * (int64_t)q], sx[3 * (int64_t)q + 1], sx[3 * (int64_t)q + 2]);
                            const int32_t j = sj[q];
                            if (d < best || (d == best && j < best_j)) { best = d; best_j = j; }
                        }
                    }
                }
            // every point of a shell beyond r is farther than r cells (minus the rounding slack of the cell assignment)
            const T reach = (T)(0.995 * (double)r) * cell;
            done = best < reach * reach;
        }
    }
    if (done) {
        d2[i] = best;
        idx[i] = best_j;
    } else {
        left[atomicAdd(n_left, 1)] = (int32_t)i;
    }
}

// the exhaustive search for the queries the grid did not finish: one workgroup per 256 of them, b through LDS tiles
template <typename T>
__global__ void grid_leftover_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t m, const int32_t* __restrict__ left,
                                     const int32_t* __restrict__ n_left, T* __restrict__ d2, int64_t* __restrict__ idx) {
    __shared__ T tx[kNnTile], ty[kNnTile], tz[kNnTile];
    const int total = *n_left;
    for (int base_q = blockIdx.x * blockDim.x; base_q < total; base_q += gridDim.x * blockDim.x) {
        const int qi = base_q + threadIdx.x;
        const int64_t i = qi < total ? left[qi] : -1;
        const T ax = i >= 0 ? a[3 * i] : (T)0, ay = i >= 0 ? a[3 * i + 1] : (T)0, az = i >= 0 ? a[3 * i + 2] : (T)0;
        T best = (T)INFINITY;
        int32_t best_j = 0;
        for (int64_t base = 0; base < m; base += kNnTile) {
            const int cnt = (int)min((int64_t)kNnTile, m - base);
            __syncthreads();
            for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
                tx[t] = b[3 * (base + t)]; ty[t] = b[3 * (base + t) + 1]; tz[t] = b[3 * (base + t) + 2];
            }
            __syncthreads();
            for (int t = 0; t < cnt; ++t) {
                const T d = nn_d2(ax, ay, az, tx[t], ty[t], tz[t]);
                if (d < best) { best = d; best_j = (int32_t)(base + t); }
            }
        }
        if (i >= 0) { d2[i] = best; idx[i] = best_j; }
    }
}

struct GridWs {            // workspace layout of lidiff_nn_dist_grid (all offsets 16-byte aligned)
    int32_t* cells; uint64_t* hkeys; int32_t* hvals; int32_t* uniq; int32_t* cell_of; int32_t* unique_ws; int32_t* start;
    int32_t* cursor; int32_t* sj; int32_t* left; int32_t* head; void* sx; int64_t cap, bytes;
};

static GridWs grid_ws(void* ws, int64_t n, int64_t m, int eb) {
    GridWs w{};
    char* p = (char*)ws;
    auto take = [&](int64_t bytes) { char* q = p; p += (bytes + 15) / 16 * 16; return q; };
    w.cap = 1024;
    while (w.cap < 2 * m) w.cap <<= 1;
    w.head = (int32_t*)take(64);                                   // [0] cells, [1] leftover queries, [2] status
    w.cells = (int32_t*)take(m * 16);
    w.hkeys = (uint64_t*)take(w.cap * 8);
    w.hvals = (int32_t*)take(w.cap * 4);
    w.uniq = (int32_t*)take(m * 16);
    w.cell_of = (int32_t*)take(m * 4);
    w.unique_ws = (int32_t*)take((m + ceil_div(m > 0 ? m : 1, kBlock) + 16) * 4);
    w.start = (int32_t*)take((m + 2) * 4);
    w.cursor = (int32_t*)take((m + 1) * 4);
    w.sj = (int32_t*)take(m * 4);
    w.left = (int32_t*)take(n * 4);
    w.sx = (void*)take(m * 3 * eb);
    w.bytes = p - (char*)ws;
    return w;
}

template <typename T>
static int nn_dist_grid_launch(const void* a, int64_t n, const void* b, int64_t m, double cell, void* d2, int64_t* idx, void* ws,
                               hipStream_t st) {
    GridWs w = grid_ws(ws, n, m, (int)sizeof(T));
    const T inv_cell = (T)(1.0 / cell);
    LIDIFF_CHECK_HIP(hipMemsetAsync(w.head, 0, 64, st));
    grid_cells_kernel<T><<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>((const T*)b, m, inv_cell, w.cells);
    const int rc = run_unique(w.cells, m, nullptr, 1, w.hkeys, w.hvals, w.cap, w.uniq, nullptr, w.cell_of, false, w.head, w.head + 2,
                              w.unique_ws, st);
    if (rc != 0) return rc;
    // cell sizes -> slice starts (over the bound m: cells behind the count are empty) -> points in cell order
    LIDIFF_CHECK_HIP(hipMemsetAsync(w.start, 0, (size_t)(m + 2) * 4, st));
    LIDIFF_CHECK_HIP(hipMemsetAsync(w.cursor, 0, (size_t)(m + 1) * 4, st));
    grid_count_kernel<<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>(w.cell_of, m, w.start);
    scan_i32_kernel<<<1, 1024, 0, st>>>(w.start, m + 1);
    grid_fill_kernel<T><<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>((const T*)b, w.cell_of, m, w.start, w.cursor, (T*)w.sx, w.sj);
    grid_query_kernel<T><<<(unsigned)ceil_div(n, kBlock), kBlock, 0, st>>>((const T*)a, n, (T)cell, inv_cell, w.hkeys, w.hvals,
                                                                          (uint32_t)(w.cap - 1), w.start, (const T*)w.sx, w.sj,
                                                                          w.head + 2, (T*)d2, idx, w.left, w.head + 1);
    grid_leftover_kernel<T><<<256, kBlock, 0, st>>>((const T*)a, (const T*)b, m, w.left, w.head + 1, (T*)d2, idx);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

static int nn_dist_splits(int64_t n, int64_t m) {
    const int64_t blocks_a = ceil_div(n > 0 ? n : 1, (int64_t)kBlock * kNnPt);
    int64_t s = ceil_div((int64_t)2048, blocks_a);                 // >= 8 workgroups per CU in flight
    s = min(s, ceil_div(m > 0 ? m : 1, (int64_t)kNnTile));
    return (int)max((int64_t)1, min(s, (int64_t)256));
}

template <typename T>
static void nn_dist_launch(const void* a, int64_t n, const void* b, int64_t m, void* d2, int64_t* idx, void* ws,
                           hipStream_t st) {
    const int splits = nn_dist_splits(n, m);
    const int64_t per = ceil_div(ceil_div(m, (int64_t)splits), (int64_t)kNnTile) * kNnTile;
    T* part_d2 = (T*)ws;
    int32_t* part_idx = (int32_t*)(part_d2 + (int64_t)splits * n);
    const dim3 grid((unsigned)ceil_div(n, (int64_t)kBlock * kNnPt), (unsigned)splits);
    nn_dist_kernel<T><<<grid, kBlock, 0, st>>>((const T*)a, n, (const T*)b, m, per, part_d2, part_idx);
    nn_dist_merge_kernel<T><<<(unsigned)ceil_div(n, kBlock), kBlock, 0, st>>>(part_d2, part_idx, n, splits, (T*)d2, idx);
}

template <bool F32IN>
static int nn_match_launch(const int32_t* full, int64_t m_full, const int32_t* part, int64_t m_part,
                           const int32_t* d_max_coord, int64_t* idx, hipStream_t st, const int32_t* d_m_full = nullptr,
                           int32_t* d_gate = nullptr) {
    // (d_m_full: m_full is only a BOUND of the row count, which the kernels read from the device)
    const int64_t blocks = ceil_div(m_full, (int64_t)kBlock * kMatchPt);
    const int64_t splits = min(ceil_div((int64_t)2048, blocks), ceil_div(m_part, (int64_t)kMatchTile));
    if (d_gate != nullptr && !F32IN && d_m_full == nullptr) {
        // by batch element first (d_gate: one device word of the caller's): every row against its own element's part rows; then
        // the unrestricted pass over the same idx[] -- its workgroups leave at once unless the check asked for it
        const int64_t sp = max((int64_t)1, splits);
        const int64_t per = ceil_div(ceil_div(m_part, sp), (int64_t)kMatchTile) * kMatchTile;
        LIDIFF_CHECK_HIP(hipMemsetAsync(idx, 0xff, (size_t)m_full * 8, st));
        LIDIFF_CHECK_HIP(hipMemsetAsync(d_gate, 0, sizeof(int32_t), st));
        nn_match_kernel<true, F32IN><<<dim3((unsigned)blocks, (unsigned)sp), kBlock, 0, st>>>(
            full, m_full, part, m_part, per, d_max_coord, idx, nullptr, nullptr, 1);
        nn_match_check_kernel<<<(unsigned)ceil_div(max(m_full, m_part), kBlock), kBlock, 0, st>>>(idx, m_full, part, m_part, d_max_coord, d_gate);
        nn_match_kernel<true, F32IN><<<dim3((unsigned)blocks, (unsigned)sp), kBlock, 0, st>>>(
            full, m_full, part, m_part, per, d_max_coord, idx, nullptr, d_gate, 0);
        nn_match_finish_kernel<<<(unsigned)ceil_div(m_full, kBlock), kBlock, 0, st>>>(idx, m_full, nullptr);
        LIDIFF_CHECK_LAUNCH();
        return 0;
    }
    if (splits <= 1) {
        nn_match_kernel<false, F32IN><<<(unsigned)blocks, kBlock, 0, st>>>(full, m_full, part, m_part, 0, d_max_coord, idx, d_m_full);
    } else {
        const int64_t per = ceil_div(ceil_div(m_part, splits), (int64_t)kMatchTile) * kMatchTile;
        LIDIFF_CHECK_HIP(hipMemsetAsync(idx, 0xff, (size_t)m_full * 8, st));
        nn_match_kernel<true, F32IN><<<dim3((unsigned)blocks, (unsigned)splits), kBlock, 0, st>>>(
            full, m_full, part, m_part, per, d_max_coord, idx, d_m_full);
        nn_match_finish_kernel<<<(unsigned)ceil_div(m_full, kBlock), kBlock, 0, st>>>(idx, m_full, d_m_full);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

}  // namespace lidiff

// =======================================================================================
using namespace lidiff;

extern "C" {

int lidiff_abi_version(void) { return LIDIFF_ABI_VERSION; }
const char* lidiff_last_error(void) { return g_err; }

int64_t lidiff_hash_capacity(int64_t n_rows) {
    int64_t cap = 1024;
    while (cap < 2 * n_rows) cap <<= 1;
    return cap;
}

int64_t lidiff_unique_workspace_bytes(int64_t n_rows) {
    return (n_rows + ceil_div(n_rows > 0 ? n_rows : 1, kBlock) + 16) * (int64_t)sizeof(int32_t);
}

int lidiff_coords_floor(const float* coords_f, int64_t n_rows, int32_t* coords_i, void* stream) {
    LIDIFF_CHECK_ARG(n_rows >= 0, "negative row count");
    if (n_rows == 0) return 0;
    const int64_t n = n_rows * 4;
    floor_kernel<<<(unsigned)ceil_div(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(coords_f, coords_i, n);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_vox_unique(const int32_t* coords, int64_t n_rows, uint64_t* hkeys, int32_t* hvals,
                      int64_t cap, int32_t* uniq, int32_t* first_idx, int64_t* inverse,
                      int32_t* d_m, int32_t* d_status, void* workspace, int32_t preinit, void* stream) {
    return run_unique(coords, n_rows, nullptr, 1, hkeys, hvals, cap, uniq, first_idx, inverse, true, d_m,
                      d_status, workspace, (hipStream_t)stream, preinit != 0);
}

int lidiff_map_stride(const int32_t* coords, int64_t n_rows, int32_t s_out, uint64_t* hkeys,
                      int32_t* hvals, int64_t cap, int32_t* coarse, int32_t* parent, int32_t* d_m,
                      int32_t* d_status, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(s_out >= 1, "stride must be >= 1");
    return run_unique(coords, n_rows, nullptr, s_out, hkeys, hvals, cap, coarse, nullptr, parent, false, d_m,
                      d_status, workspace, (hipStream_t)stream);
}

int lidiff_map_stride_dev(const int32_t* coords, int64_t n_rows_bound, const int32_t* d_n_rows, int32_t s_out,
                          uint64_t* hkeys, int32_t* hvals, int64_t cap, int32_t* coarse, int32_t* parent, int32_t* d_m,
                          int32_t* d_status, void* workspace, int32_t preinit, void* stream) {
    LIDIFF_CHECK_ARG(s_out >= 1, "stride must be >= 1");
    LIDIFF_CHECK_ARG(d_n_rows != nullptr, "d_n_rows");
    return run_unique(coords, n_rows_bound, d_n_rows, s_out, hkeys, hvals, cap, coarse, nullptr, parent, false, d_m,
                      d_status, workspace, (hipStream_t)stream, preinit != 0);
}

int64_t lidiff_vox_mean_workspace_bytes(int64_t m, int32_t c) {
    return (m < 0 || c <= 0) ? 0 : 16 + m * (int64_t)c * 8 + ((m * 4 + 15) / 16) * 16;
}

int lidiff_vox_mean(const float* feats, const int64_t* inverse, int64_t n_rows, int32_t c, int64_t m,
                    float* out, float* counts, void* workspace, int32_t preinit, void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && m >= 0 && n_rows >= 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (m == 0) return 0;
    LIDIFF_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
    // workspace: [max |x| bits (16 B)] [fixed-point sums m x c int64] [point counts m int32]
    uint32_t* amax = (uint32_t*)workspace;
    unsigned long long* acc = (unsigned long long*)((char*)workspace + 16);
    int32_t* cnt = (int32_t*)(acc + m * c);
    if (!preinit) LIDIFF_CHECK_HIP(hipMemsetAsync(workspace, 0, (size_t)lidiff_vox_mean_workspace_bytes(m, c), st));
    int n_bits = 1;
    while ((1ll << n_bits) <= n_rows) ++n_bits;                       // n_rows < 2^n_bits
    if (n_rows > 0) {
        const int64_t total = n_rows * c;
        mean_absmax_kernel<<<(unsigned)(ceil_div(total, 4 * kBlock) < 256 ? ceil_div(total, 4 * kBlock) : 256), kBlock, 0, st>>>(feats, total, amax);
        mean_accum_kernel<<<(unsigned)ceil_div(n_rows, kBlock), kBlock, 0, st>>>(feats, inverse, n_rows, c, n_bits, amax, acc, cnt);
    }
    mean_div_kernel<<<(unsigned)ceil_div(m * c, kBlock), kBlock, 0, st>>>(acc, cnt, m * c, c, n_bits, amax, feats, inverse, n_rows, out, counts);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_vox_mean_bwd(const float* grad_out, const int64_t* inverse, const float* counts,
                        int64_t n_rows, int32_t c, float* grad_feats, void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && n_rows >= 0, "bad shape");
    if (n_rows == 0) return 0;
    mean_bwd_kernel<<<(unsigned)ceil_div(n_rows * c, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        grad_out, inverse, counts, n_rows * c, c, grad_feats);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map(const int32_t* out_coords, int64_t m_out, const uint64_t* hkeys_in,
                      const int32_t* hvals_in, int64_t cap_in, int32_t ks, int32_t step, int32_t* nbr,
                      void* stream) {
    LIDIFF_CHECK_ARG(ks >= 1 && ks <= 3, "kernel_size must be 1..3");
    LIDIFF_CHECK_ARG(cap_in > 0 && (cap_in & (cap_in - 1)) == 0, "cap must be a power of two");
    if (m_out == 0) return 0;
    const unsigned blocks = (unsigned)ceil_div(m_out, kBlock);
    const uint32_t mask = (uint32_t)(cap_in - 1);
    hipStream_t st = (hipStream_t)stream;
    if (ks == 3) kernel_map_kernel<3><<<blocks, kBlock, 0, st>>>(out_coords, m_out, hkeys_in, hvals_in, mask, step, nbr);
    else if (ks == 2) kernel_map_kernel<2><<<blocks, kBlock, 0, st>>>(out_coords, m_out, hkeys_in, hvals_in, mask, step, nbr);
    else kernel_map_kernel<1><<<blocks, kBlock, 0, st>>>(out_coords, m_out, hkeys_in, hvals_in, mask, step, nbr);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_self(const int32_t* coords, int64_t m, const uint64_t* hkeys, const int32_t* hvals, int64_t cap,
                           int32_t step, int32_t* nbr, void* stream) {
    LIDIFF_CHECK_ARG(cap > 0 && (cap & (cap - 1)) == 0, "cap must be a power of two");
    LIDIFF_CHECK_ARG(step >= 1, "step must be >= 1");
    if (m == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    LIDIFF_CHECK_HIP(hipMemsetAsync(nbr, 0xff, (size_t)27 * m * sizeof(int32_t), st));
    kernel_map_self_kernel<<<(unsigned)ceil_div(m, kBlock), kBlock, 0, st>>>(coords, m, nullptr, hkeys, hvals,
                                                                             (uint32_t)(cap - 1), step, nbr);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_self_dev(const int32_t* coords, int64_t m_bound, const int32_t* d_m, const uint64_t* hkeys,
                               const int32_t* hvals, int64_t cap, int32_t step, int32_t* nbr, int32_t preinit, void* stream) {
    LIDIFF_CHECK_ARG(cap > 0 && (cap & (cap - 1)) == 0, "cap must be a power of two");
    LIDIFF_CHECK_ARG(step >= 1 && d_m != nullptr, "step must be >= 1, d_m set");
    if (m_bound == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!preinit) LIDIFF_CHECK_HIP(hipMemsetAsync(nbr, 0xff, (size_t)27 * m_bound * sizeof(int32_t), st));
    kernel_map_self_kernel<<<(unsigned)ceil_div(m_bound, kBlock), kBlock, 0, st>>>(coords, m_bound, d_m, hkeys, hvals,
                                                                                   (uint32_t)(cap - 1), step, nbr);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_down(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine, int32_t ts_fine,
                           int64_t m_coarse, int32_t* nbr_down, void* stream) {
    LIDIFF_CHECK_ARG(ts_fine >= 1, "tensor stride must be >= 1");
    if (m_coarse == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    LIDIFF_CHECK_HIP(hipMemsetAsync(nbr_down, 0xff, (size_t)8 * m_coarse * sizeof(int32_t), st));
    if (m_fine == 0) return 0;
    kernel_map_down_kernel<<<(unsigned)ceil_div(m_fine, kBlock), kBlock, 0, st>>>(fine_coords, parent, m_fine, nullptr, ts_fine,
                                                                                  m_coarse, nbr_down);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_down_dev(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine_bound, const int32_t* d_m_fine,
                               int32_t ts_fine, int64_t m_coarse_bound, int32_t* nbr_down, int32_t preinit, void* stream) {
    LIDIFF_CHECK_ARG(ts_fine >= 1 && d_m_fine != nullptr, "tensor stride must be >= 1, d_m_fine set");
    if (m_coarse_bound == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!preinit) LIDIFF_CHECK_HIP(hipMemsetAsync(nbr_down, 0xff, (size_t)8 * m_coarse_bound * sizeof(int32_t), st));
    if (m_fine_bound == 0) return 0;
    kernel_map_down_kernel<<<(unsigned)ceil_div(m_fine_bound, kBlock), kBlock, 0, st>>>(fine_coords, parent, m_fine_bound, d_m_fine,
                                                                                        ts_fine, m_coarse_bound, nbr_down);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_up(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine,
                         int32_t ts_fine, int32_t* nbr_up, void* stream) {
    LIDIFF_CHECK_ARG(ts_fine >= 1, "tensor stride must be >= 1");
    if (m_fine == 0) return 0;
    kernel_map_up_kernel<<<(unsigned)ceil_div(m_fine, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        fine_coords, parent, m_fine, nullptr, ts_fine, nbr_up);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_kernel_map_up_dev(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine_bound, const int32_t* d_m_fine,
                             int32_t ts_fine, int32_t* nbr_up, void* stream) {
    LIDIFF_CHECK_ARG(ts_fine >= 1 && d_m_fine != nullptr, "tensor stride must be >= 1, d_m_fine set");
    if (m_fine_bound == 0) return 0;
    kernel_map_up_kernel<<<(unsigned)ceil_div(m_fine_bound, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        fine_coords, parent, m_fine_bound, d_m_fine, ts_fine, nbr_up);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_morton_keys(const int32_t* coords, int64_t m, int32_t ts, int64_t* keys, void* stream) {
    LIDIFF_CHECK_ARG(ts >= 1 && m >= 0, "bad shape");
    if (m == 0) return 0;
    morton_kernel<<<(unsigned)ceil_div(m, kBlock), kBlock, 0, (hipStream_t)stream>>>(coords, m, ts, keys);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int64_t lidiff_rulebook_workspace_bytes(int32_t k_vol, int64_t m_out) {
    return ((int64_t)k_vol * ceil_div(m_out > 0 ? m_out : 1, kBlock) + 16) * (int64_t)sizeof(int32_t);
}

int lidiff_rulebook_compact(const int32_t* nbr, int32_t k_vol, int64_t m_out, int32_t* offset_ptr,
                            int32_t* pairs_in, int32_t* pairs_out, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(k_vol >= 1 && m_out >= 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (m_out == 0) {
        LIDIFF_CHECK_HIP(hipMemsetAsync(offset_ptr, 0, (size_t)(k_vol + 1) * sizeof(int32_t), st));
        return 0;
    }
    const int nblk = (int)ceil_div(m_out, kBlock);
    int32_t* counts = (int32_t*)workspace;
    rb_count_kernel<<<dim3(nblk, k_vol), kBlock, 0, st>>>(nbr, m_out, counts);
    rb_scan_kernel<<<1, 1024, 0, st>>>(counts, nblk * k_vol, nblk, k_vol, offset_ptr);
    if (pairs_in != nullptr)
        rb_fill_kernel<<<dim3(nblk, k_vol), kBlock, 0, st>>>(nbr, m_out, counts, pairs_in, pairs_out);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int64_t lidiff_tail_map_workspace_bytes(int32_t k_vol, int64_t m_out) {
    return ((int64_t)(k_vol + 1) * ceil_div(m_out > 0 ? m_out : 1, kBlock) + 32) * (int64_t)sizeof(int32_t);
}

static int tail_map_impl(const int32_t* nbr, int32_t k_vol, int64_t m_out, const int32_t* d_m, int32_t skip,
                         int32_t* offset_ptr, int32_t* row_ptr, int64_t n_pairs, int32_t* tail_nbr, int32_t* idx,
                         void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && m_out >= 0, "bad shape");
    LIDIFF_CHECK_ARG(nbr != nullptr && offset_ptr != nullptr && row_ptr != nullptr && workspace != nullptr, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)ceil_div(m_out > 0 ? m_out : 1, kBlock);
    int32_t* counts = (int32_t*)workspace;
    if (tail_nbr == nullptr) {                       // phase 1: counts, offset_ptr[k_vol + 1], row_ptr[m_out + 1]
        if (m_out == 0) {
            LIDIFF_CHECK_HIP(hipMemsetAsync(offset_ptr, 0, (size_t)(k_vol + 1) * sizeof(int32_t), st));
            LIDIFF_CHECK_HIP(hipMemsetAsync(row_ptr, 0, sizeof(int32_t), st));
            return 0;
        }
        tail_count_kernel<<<dim3(nblk, k_vol), kBlock, 0, st>>>(nbr, m_out, d_m, skip, counts);
        rb_scan_kernel<<<1, 1024, 0, st>>>(counts, nblk * k_vol, nblk, k_vol, offset_ptr);
        int32_t* blk_sums = counts + (int64_t)k_vol * nblk + 8;       // behind the per-offset block counts
        tail_rowcnt_kernel<false><<<nblk, kBlock, 0, st>>>(nbr, m_out, d_m, k_vol, skip, blk_sums, row_ptr);
        scan_i32_kernel<<<1, 1024, 0, st>>>(blk_sums, nblk);
        tail_rowcnt_kernel<true><<<nblk, kBlock, 0, st>>>(nbr, m_out, d_m, k_vol, skip, blk_sums, row_ptr);
        LIDIFF_CHECK_LAUNCH();
        return 0;
    }
    LIDIFF_CHECK_ARG(idx != nullptr && n_pairs >= 0, "phase 2 needs tail_nbr, idx and the pair count of phase 1");
    if (n_pairs == 0 || m_out == 0) return 0;
    LIDIFF_CHECK_HIP(hipMemsetAsync(tail_nbr, 0xFF, (size_t)k_vol * n_pairs * sizeof(int32_t), st));
    tail_fill_kernel<<<dim3(nblk, k_vol), kBlock, 0, st>>>(nbr, m_out, d_m, skip, n_pairs, counts, row_ptr, tail_nbr, idx);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_tail_map(const int32_t* nbr, int32_t k_vol, int64_t m_out, int32_t skip, int32_t* offset_ptr,
                    int32_t* row_ptr, int64_t n_pairs, int32_t* tail_nbr, int32_t* idx, void* workspace, void* stream) {
    return tail_map_impl(nbr, k_vol, m_out, nullptr, skip, offset_ptr, row_ptr, n_pairs, tail_nbr, idx, workspace, stream);
}

int lidiff_tail_map_dev(const int32_t* nbr, int32_t k_vol, int64_t m_bound, const int32_t* d_m, int32_t skip,
                        int32_t* offset_ptr, int32_t* row_ptr, int64_t n_pairs, int32_t* tail_nbr, int32_t* idx,
                        void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(d_m != nullptr, "d_m");
    return tail_map_impl(nbr, k_vol, m_bound, d_m, skip, offset_ptr, row_ptr, n_pairs, tail_nbr, idx, workspace, stream);
}

int lidiff_tail_map_fill_bounded(const int32_t* nbr, int32_t k_vol, int64_t m_bound, const int32_t* d_m, int32_t skip,
                                 const int32_t* offset_ptr, const int32_t* row_ptr, int64_t n_pairs_bound, int32_t* pair_in,
                                 int32_t* idx, int32_t* d_status, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(k_vol >= 1 && k_vol <= 27 && m_bound >= 0 && n_pairs_bound >= 0, "bad shape");
    LIDIFF_CHECK_ARG(nbr && offset_ptr && row_ptr && workspace && d_status, "null pointer");
    if (n_pairs_bound == 0 || m_bound == 0) return 0;
    LIDIFF_CHECK_ARG(pair_in != nullptr && idx != nullptr, "pair_in / idx");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = (int)ceil_div(m_bound, kBlock);
    LIDIFF_CHECK_HIP(hipMemsetAsync(pair_in, 0, (size_t)n_pairs_bound * sizeof(int32_t), st));
    LIDIFF_CHECK_HIP(hipMemsetAsync(idx, 0, (size_t)n_pairs_bound * sizeof(int32_t), st));
    tail_fill_bounded_kernel<<<dim3(nblk, k_vol), kBlock, 0, st>>>(nbr, m_bound, d_m, skip, k_vol, n_pairs_bound,
                                                                   (const int32_t*)workspace, offset_ptr, row_ptr, pair_in, idx,
                                                                   d_status);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_host_device_pointer(void* host_ptr, void** dev_ptr) {
    LIDIFF_CHECK_ARG(host_ptr != nullptr && dev_ptr != nullptr, "null pointer");
    LIDIFF_CHECK_HIP(hipHostGetDevicePointer(dev_ptr, host_ptr, 0));
    return 0;
}

int lidiff_publish_words(const int32_t* words, int32_t n_words, const int32_t* d_status, int32_t* host_mapped, int32_t seq,
                         void* stream) {
    LIDIFF_CHECK_ARG(words != nullptr && host_mapped != nullptr && n_words >= 0 && n_words <= 62, "up to 62 words");
    publish_kernel<<<1, 64, 0, (hipStream_t)stream>>>(words, n_words, d_status, host_mapped, seq);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_gather_rows(const float* src, const int64_t* idx, int64_t n_rows, int32_t c, float* dst,
                       void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && n_rows >= 0, "bad shape");
    if (n_rows == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    if (vec)
        gather_rows_kernel<true><<<(unsigned)ceil_div(n_rows * (c / 4), kBlock), kBlock, 0, st>>>(src, idx, n_rows, c, dst);
    else
        gather_rows_kernel<false><<<(unsigned)ceil_div(n_rows * c, kBlock), kBlock, 0, st>>>(src, idx, n_rows, c, dst);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_gather_bias_leaky(const float* src, const int64_t* idx, const float* bias, int64_t n_rows, int32_t c,
                             float slope, float* dst, void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && c % 4 == 0 && n_rows >= 0, "c must be a positive multiple of 4");
    LIDIFF_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst | (uintptr_t)bias) & 15) == 0, "pointers must be 16-byte aligned");
    if (n_rows == 0) return 0;
    gather_bias_leaky_kernel<<<(unsigned)ceil_div(n_rows * (c / 4), kBlock), kBlock, 0, (hipStream_t)stream>>>(
        src, idx, bias, n_rows, c / 4, slope, dst);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_gather_mul_rows(const float* x, const float* table, const int64_t* idx, int64_t n_rows, int32_t c,
                           float* dst, const int32_t* d_n_rows, void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && c % 4 == 0 && n_rows >= 0, "c must be a positive multiple of 4");
    LIDIFF_CHECK_ARG((((uintptr_t)x | (uintptr_t)dst | (uintptr_t)table) & 15) == 0, "pointers must be 16-byte aligned");
    if (n_rows == 0) return 0;
    gather_mul_rows_kernel<<<(unsigned)ceil_div(n_rows * (c / 4), kBlock), kBlock, 0, (hipStream_t)stream>>>(
        x, table, idx, n_rows, d_n_rows, c / 4, dst);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int64_t lidiff_segment_sum_workspace_bytes(int64_t n_sources, int32_t c) {
    return (4 + 3 * seg_cap_long(n_sources) + 2 * seg_cap_chunks(n_sources) + 8) * (int64_t)sizeof(int32_t)
           + seg_cap_chunks(n_sources) * (int64_t)c * (int64_t)sizeof(float) + 16;
}

int lidiff_segment_sum_rows(const float* src, const int64_t* order, const int64_t* ptr, int64_t m, int32_t c, float* dst,
                            int64_t n_sources, void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(c > 0 && m >= 0 && n_sources >= 0, "bad shape");
    if (m == 0) return 0;
    LIDIFF_CHECK_ARG(src && order && ptr && dst, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (c % 4 == 0) && (((uintptr_t)src | (uintptr_t)dst) % 16 == 0);
    SegWork w{};
    if (workspace != nullptr) {
        w = seg_work(workspace, n_sources, c);
        LIDIFF_CHECK_HIP(hipMemsetAsync(w.head, 0, 4 * sizeof(int32_t), st));
    }
    if (vec) segment_sum_rows_kernel<true><<<(unsigned)ceil_div(m * (c / 4), kBlock), kBlock, 0, st>>>(src, order, ptr, m, c, dst, w);
    else segment_sum_rows_kernel<false><<<(unsigned)ceil_div(m * c, kBlock), kBlock, 0, st>>>(src, order, ptr, m, c, dst, w);
    if (workspace != nullptr) {
        segment_sum_chunk_kernel<<<dim3((unsigned)min((int64_t)512, w.cap_chunks), (unsigned)ceil_div(c, 32)), 256, 0, st>>>(src, order, ptr, c, w);
        segment_sum_combine_kernel<<<(unsigned)min((int64_t)256, ceil_div(w.cap_long * c, kBlock)), kBlock, 0, st>>>(c, dst, w);
    }
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int lidiff_nn_match(const int32_t* full, int64_t m_full, const int32_t* part, int64_t m_part,
                    const int32_t* d_max_coord, int64_t* idx, int32_t* d_gate, void* stream) {
    LIDIFF_CHECK_ARG(m_part >= 1, "part tensor has no rows");
    if (m_full == 0) return 0;
    LIDIFF_CHECK_ARG(m_part < (int64_t)1 << 31, "part tensor too large for 32-bit row indices");
    return nn_match_launch<false>(full, m_full, part, m_part, d_max_coord, idx, (hipStream_t)stream, nullptr, d_gate);
}

int lidiff_nn_match_dev(const int32_t* full, int64_t m_full_bound, const int32_t* d_m_full, const int32_t* part, int64_t m_part,
                        int32_t* d_max_coord, int64_t* idx, void* stream) {
    LIDIFF_CHECK_ARG(full && d_m_full && part && d_max_coord && idx, "null pointer");
    LIDIFF_CHECK_ARG(m_part >= 1 && m_part < (int64_t)1 << 31, "the part tensor must have 1 .. 2^31-1 rows");
    if (m_full_bound == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    LIDIFF_CHECK_HIP(hipMemsetAsync(d_max_coord, 0x80, sizeof(int32_t), st));       // 0x80808080: below every coordinate
    coord_max_dev_kernel<<<(unsigned)min((int64_t)128, ceil_div(m_full_bound, 4 * kBlock)), kBlock, 0, st>>>(full, d_m_full, d_max_coord);
    return nn_match_launch<false>(full, m_full_bound, part, m_part, d_max_coord, idx, st, d_m_full);
}

int lidiff_argmin_rows_f32(const float* a, int64_t n, const float* b, int64_t m, int64_t* idx, void* stream) {
    LIDIFF_CHECK_ARG(m >= 1 && m < (int64_t)1 << 31, "the searched rows must number 1 .. 2^31-1");
    LIDIFF_CHECK_ARG((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "rows must be 16-byte aligned float4");
    if (n == 0) return 0;
    return nn_match_launch<true>((const int32_t*)a, n, (const int32_t*)b, m, nullptr, idx, (hipStream_t)stream);
}

int64_t lidiff_fps_workspace_bytes(int64_t n_points) {
    const int64_t blocks = ceil_div(n_points > 0 ? n_points : 1, kFpsBlock);
    return n_points * 8 + blocks * 16 + 64 + 4096 * 16;      // + the cooperative kernel's double-buffered partial maxima
}

int lidiff_fps(const double* points, int64_t n_points, int64_t n_samples, int64_t* selected, void* workspace,
               void* stream) {
    LIDIFF_CHECK_ARG(n_points >= 1 && n_samples >= 1 && n_samples <= n_points, "need 1 <= n_samples <= n_points");
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)ceil_div(n_points, kFpsBlock);
    double* dist = (double*)workspace;
    double* blk_val = dist + n_points;
    int64_t* blk_idx = (int64_t*)(blk_val + blocks);
    unsigned int* counter = (unsigned int*)(blk_idx + blocks);
    // distances start "infinite": bytes 0x7f give 1.4e306, above any squared distance of finite float32 coordinates
    LIDIFF_CHECK_HIP(hipMemsetAsync(dist, 0x7f, (size_t)n_points * 8, st));
    LIDIFF_CHECK_HIP(hipMemsetAsync(counter, 0, 64, st));
    LIDIFF_CHECK_HIP(hipMemsetAsync(selected, 0, 8, st));                             // selection 0 = point 0
    for (int64_t i = 0; i + 1 < n_samples; ++i)
        fps_step_kernel<<<blocks, kFpsBlock, 0, st>>>(points, dist, n_points, selected, i, blk_val, blk_idx, counter);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

// one cooperative launch (fps_coop_kernel); workspace: lidiff_fps_workspace_bytes.  *status (device int, zeroed here) becomes
// non-zero if the device-wide barrier timed out -- the selection is then incomplete and lidiff_fps must be used instead.
// Returns non-zero (nothing enqueued) if the device cannot hold the grid co-resident or n_points exceeds grid x 8192.
// grid / points-per-thread of the cooperative kernel for n_points, or false when this device cannot run it that way
static bool fps_coop_plan(int64_t n_points, int* grid_out, int* pt_out) {
    int dev = 0, cus = 0, coop = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return false;
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) != hipSuccess || !coop) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_coop_kernel, kFpsBlock, 0) != hipSuccess || per_cu < 1)
        return false;
    // as FEW workgroups as the registers allow (kFpsCoopPt points per thread): a selection costs one device-wide barrier, and a
    // barrier among 15 workgroups (119 035 points) is far cheaper than one among 256 -- the arithmetic per selection is ~100 cycles
    // per thread either way (round 4; all_cus: one workgroup per compute unit, the round-3 plan)
    const int64_t blocks = ceil_div(n_points, kFpsBlock);
    constexpr bool all_cus = false;
    const int64_t fewest = ceil_div(n_points, (int64_t)kFpsBlock * kFpsCoopPt);
    const int grid = (int)(all_cus ? (blocks < cus ? blocks : cus) : (fewest < cus ? fewest : cus));
    const int64_t pt = ceil_div(n_points, (int64_t)grid * kFpsBlock);
    if (pt > kFpsCoopPt) return false;
    *grid_out = grid;
    *pt_out = (int)pt;
    return true;
}

int32_t lidiff_fps_coop_supported(int64_t n_points) {
    int grid = 0, pt = 0;
    return n_points >= 1 && fps_coop_plan(n_points, &grid, &pt) ? 1 : 0;
}

int lidiff_fps_coop(const double* points, int64_t n_points, int64_t n_samples, int64_t* selected, void* workspace,
                    int32_t* status, void* stream) {
    LIDIFF_CHECK_ARG(n_points >= 1 && n_samples >= 1 && n_samples <= n_points, "need 1 <= n_samples <= n_points");
    LIDIFF_CHECK_ARG(status != nullptr && workspace != nullptr, "null pointer");
    hipStream_t st = (hipStream_t)stream;
    int grid = 0, pt = 0;
    LIDIFF_CHECK_ARG(fps_coop_plan(n_points, &grid, &pt),
                     "no cooperative launch for this device / point count (ask lidiff_fps_coop_supported first)");
    double* part_val = (double*)workspace;
    int64_t* part_idx = (int64_t*)(part_val + 2 * grid);
    unsigned int* counter = (unsigned int*)(part_idx + 2 * grid);
    LIDIFF_CHECK_ARG(2 * grid * 16 + 64 <= lidiff_fps_workspace_bytes(n_points), "workspace too small");
    LIDIFF_CHECK_HIP(hipMemsetAsync(counter, 0, 64, st));
    LIDIFF_CHECK_HIP(hipMemsetAsync(status, 0, 4, st));
    void* args[] = {(void*)&points, (void*)&n_points, (void*)&n_samples, (void*)&pt, (void*)&selected, (void*)&part_val,
                    (void*)&part_idx, (void*)&counter, (void*)&status};
    LIDIFF_CHECK_HIP(hipLaunchCooperativeKernel((const void*)fps_coop_kernel, dim3(grid), dim3(kFpsBlock), args, 0, st));
    return 0;
}

int64_t lidiff_nn_dist_workspace_bytes(int64_t n, int64_t m, int32_t elem_bytes) {
    return (int64_t)nn_dist_splits(n, m) * (n > 0 ? n : 1) * (elem_bytes + 4) + 64;
}

int lidiff_nn_dist(const void* a, int64_t n, const void* b, int64_t m, int32_t elem_bytes, void* d2, int64_t* idx,
                   void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "elem_bytes must be 4 (float) or 8 (double)");
    LIDIFF_CHECK_ARG(m >= 1, "the searched cloud has no points");
    LIDIFF_CHECK_ARG(m < (int64_t)1 << 31, "the searched cloud is too large for 32-bit row indices");
    if (n == 0) return 0;
    if (elem_bytes == 4) nn_dist_launch<float>(a, n, b, m, d2, idx, workspace, (hipStream_t)stream);
    else nn_dist_launch<double>(a, n, b, m, d2, idx, workspace, (hipStream_t)stream);
    LIDIFF_CHECK_LAUNCH();
    return 0;
}

int64_t lidiff_nn_dist_grid_workspace_bytes(int64_t n, int64_t m, int32_t elem_bytes) {
    return grid_ws(nullptr, n > 0 ? n : 1, m > 0 ? m : 1, elem_bytes).bytes + 64;
}

int lidiff_nn_dist_grid(const void* a, int64_t n, const void* b, int64_t m, int32_t elem_bytes, double cell, void* d2, int64_t* idx,
                        void* workspace, void* stream) {
    LIDIFF_CHECK_ARG(elem_bytes == 4 || elem_bytes == 8, "elem_bytes must be 4 (float) or 8 (double)");
    LIDIFF_CHECK_ARG(m >= 1 && m < (int64_t)1 << 30 && n < (int64_t)1 << 31, "cloud sizes");
    LIDIFF_CHECK_ARG(cell > 0.0, "cell edge must be positive");
    LIDIFF_CHECK_ARG(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "workspace must be 16-byte aligned");
    if (n == 0) return 0;
    if (elem_bytes == 4) return nn_dist_grid_launch<float>(a, n, b, m, cell, d2, idx, workspace, (hipStream_t)stream);
    return nn_dist_grid_launch<double>(a, n, b, m, cell, d2, idx, workspace, (hipStream_t)stream);
}

}  // extern "C"
