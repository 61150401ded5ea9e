// Shared device/host helpers for the lidiff_amd HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/lidiff_amd.h"

namespace lidiff {

constexpr uint64_t kEmptyKey = ~0ull;          // hipMemset(0xFF) initialises the table
constexpr int32_t kKeyOff = 32768;
constexpr int kWave = 64;                      // gfx950 wavefront

void set_error(const char* fmt, ...);

#define LIDIFF_CHECK_ARG(cond, msg)                         \
    do {                                                    \
        if (!(cond)) {                                      \
            ::lidiff::set_error("%s: %s", __func__, msg);   \
            return 1;                                       \
        }                                                   \
    } while (0)

#define LIDIFF_CHECK_HIP(expr)                                                        \
    do {                                                                              \
        hipError_t e__ = (expr);                                                      \
        if (e__ != hipSuccess) {                                                      \
            ::lidiff::set_error("%s: %s -> %s", __func__, #expr, hipGetErrorString(e__)); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

#define LIDIFF_CHECK_LAUNCH() LIDIFF_CHECK_HIP(hipGetLastError())

// 16 bits per column, offset binary.  ok=false if any column leaves the range.
__device__ __forceinline__ uint64_t pack_key(int b, int x, int y, int z, bool& ok) {
    const unsigned ub = (unsigned)(b + kKeyOff), ux = (unsigned)(x + kKeyOff),
                   uy = (unsigned)(y + kKeyOff), uz = (unsigned)(z + kKeyOff);
    const uint64_t key = ((uint64_t)ub << 48) | ((uint64_t)ux << 32) | ((uint64_t)uy << 16) | (uint64_t)uz;
    ok = ((ub | ux | uy | uz) >> 16) == 0 && key != kEmptyKey;      // (32767,32767,32767,32767) is the empty marker
    return key;
}

// murmur3 fmix64
__device__ __forceinline__ uint32_t hash_key(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return (uint32_t)k;
}

// floor division for any s > 0
__device__ __forceinline__ int floor_div(int a, int s) {
    int q = a / s;
    return (a % s != 0 && ((a < 0) != (s < 0))) ? q - 1 : q;
}

// lookup: row stored for `key`, or -1
__device__ __forceinline__ int hash_find(const uint64_t* __restrict__ hkeys,
                                         const int32_t* __restrict__ hvals, uint32_t mask,
                                         uint64_t key) {
    uint32_t slot = hash_key(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint64_t k = hkeys[slot];
        if (k == key) return hvals[slot];
        if (k == kEmptyKey) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int popc_below(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace lidiff
