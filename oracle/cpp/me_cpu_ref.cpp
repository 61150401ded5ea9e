// C++17 / OpenMP restatement of MinkowskiEngine 0.5.4's CPU algorithm for the operators LiDiff consumes.
//
// TEST INFRASTRUCTURE (see oracle/me_cpu.py): the checker's fast leg and the "CPU path" timed beside the GPU numbers
// (bench.py cpu_baseline).  PARITY UNPINNED: MinkowskiEngine is not installable here; this file
// restates its published CPU algorithm, exactly as oracle/me_cpu.py does in numpy -- the two are
// cross-checked bit for bit (maps) / to 1e-5 (features) by tests/test_oracle.py:
//   * coordinate map = hash map from the packed (b, x, y, z) key to the row id, rows in first-occurrence order
//     (ME CPU: CoordinateMapCPU::insert_and_map, robin-hood flat map; A.3 / A.4);
//   * strided map = floor(c / s) * s re-inserted in fine-row order (CoordinateMapManager::stride, A.5);
//   * kernel map = per kernel offset a probe of the input map for every output coordinate (kernel_map, A.6), kept as the
//     neighbour table nbr[K][M_out] (input row or -1), equivalent to ME's per-offset in/out lists sorted by output row;
//   * convolution forward = for k ascending: gather the offset's input rows into a dense buffer, SGEMM with W[k] (MKL,
//     the BLAS torch itself links), scatter-add the result rows (ConvolutionForwardKernelCPU, A.6): fp32, sum order =
//     ascending k;
//   * arg-min match = exact integer squared distance over (b * 2 max, x, y, z), lowest index on ties (A.9).
// Call sites in the reference: lidiff/models/minkunet.py:13-80 (blocks), :403-418 (match); tools/diff_completion_pipeline.py:68-84,149.
#include <omp.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

extern "C" void sgemm_(const char* transa, const char* transb, const int* m, const int* n, const int* k, const float* alpha,
                       const float* a, const int* lda, const float* b, const int* ldb, const float* beta, float* c,
                       const int* ldc);

namespace {

constexpr int64_t kOff = 32768;
inline uint64_t pack(const int32_t* c) {
    return ((uint64_t)(c[0] + kOff) << 48) | ((uint64_t)(c[1] + kOff) << 32) | ((uint64_t)(c[2] + kOff) << 16) |
           (uint64_t)(c[3] + kOff);
}
inline bool in_range(const int32_t* c) {
    for (int i = 0; i < 4; ++i)
        if (c[i] < -kOff || c[i] >= kOff) return false;
    return !(c[0] == kOff - 1 && c[1] == kOff - 1 && c[2] == kOff - 1 && c[3] == kOff - 1);
}
inline uint64_t mix(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}

struct Map {                                   // open addressing, linear probing
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    uint64_t mask = 0;
    void reserve(int64_t n) {
        uint64_t cap = 1024;
        while (cap < 2 * (uint64_t)n) cap <<= 1;
        keys.assign(cap, ~0ull);
        vals.assign(cap, -1);
        mask = cap - 1;
    }
    int32_t insert(uint64_t key, int32_t row) {    // returns the row stored for key (row if it was new)
        uint64_t s = mix(key) & mask;
        for (;;) {
            if (keys[s] == ~0ull) { keys[s] = key; vals[s] = row; return row; }
            if (keys[s] == key) return vals[s];
            s = (s + 1) & mask;
        }
    }
    int32_t find(uint64_t key) const {
        uint64_t s = mix(key) & mask;
        for (;;) {
            if (keys[s] == key) return vals[s];
            if (keys[s] == ~0ull) return -1;
            s = (s + 1) & mask;
        }
    }
};

inline int32_t floor_div(int32_t a, int32_t s) {
    int32_t q = a / s;
    return (a % s != 0 && ((a < 0) != (s < 0))) ? q - 1 : q;
}

}  // namespace

extern "C" {

int me_ref_threads() { return omp_get_max_threads(); }
void me_ref_set_threads(int n) { omp_set_num_threads(n); }

// unique rows in first-occurrence order after flooring columns 1..3 to multiples of s (s = 1: voxelize).
// uniq [n,4] (first *m valid), inverse [n] (int64), first_idx [n].  Returns 0, or 1 if a coordinate leaves the key range.
int me_ref_unique(const int32_t* coords, int64_t n, int32_t s, int32_t* uniq, int64_t* inverse, int32_t* first_idx,
                  int64_t* m_out) {
    Map map;
    map.reserve(n);
    int32_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t c[4] = {coords[4 * i], coords[4 * i + 1], coords[4 * i + 2], coords[4 * i + 3]};
        if (s > 1)
            for (int d = 1; d < 4; ++d) c[d] = floor_div(c[d], s) * s;
        if (!in_range(c)) return 1;
        const int32_t row = map.insert(pack(c), m);
        if (row == m) {
            memcpy(uniq + 4 * (int64_t)m, c, 16);
            first_idx[m] = (int32_t)i;
            ++m;
        }
        inverse[i] = row;
    }
    *m_out = m;
    return 0;
}

// nbr[k * m_out + o] = row of in_coords equal to out_coords[o] + offset_k * step (x fastest; odd ks centred, even ks {0, 1})
void me_ref_kernel_map(const int32_t* in_coords, int64_t m_in, const int32_t* out_coords, int64_t m_out, int32_t ks,
                       int32_t step, int32_t* nbr) {
    Map map;
    map.reserve(m_in);
    for (int64_t i = 0; i < m_in; ++i) map.insert(pack(in_coords + 4 * i), (int32_t)i);
    const int lo = (ks & 1) ? -(ks - 1) / 2 : 0;
    const int kvol = ks * ks * ks;
#pragma omp parallel for schedule(static)
    for (int64_t o = 0; o < m_out; ++o) {
        const int32_t* c = out_coords + 4 * o;
        int k = 0;
        for (int dz = 0; dz < ks; ++dz)
            for (int dy = 0; dy < ks; ++dy)
                for (int dx = 0; dx < ks; ++dx, ++k) {
                    int32_t q[4] = {c[0], c[1] + (dx + lo) * step, c[2] + (dy + lo) * step, c[3] + (dz + lo) * step};
                    nbr[(int64_t)k * m_out + o] = in_range(q) ? map.find(pack(q)) : -1;
                }
        (void)kvol;
    }
}

// out [m_out, c_out] = sum over k ascending of gather(in)[pairs of k] @ w[k]   (w [K, c_in, c_out] row-major)
void me_ref_conv_forward(const float* in, int64_t m_in, int32_t c_in, const float* w, int32_t k_vol, int32_t c_out,
                         const int32_t* nbr, int64_t m_out, float* out) {
    memset(out, 0, sizeof(float) * (size_t)m_out * c_out);
    std::vector<int32_t> rows_in, rows_out;
    std::vector<float> buf, res;
    for (int k = 0; k < k_vol; ++k) {
        rows_in.clear();
        rows_out.clear();
        if (nbr == nullptr) {                    // kernel_size 1: F.mm(kernel)
            rows_in.resize(m_out);
            rows_out.resize(m_out);
            for (int64_t o = 0; o < m_out; ++o) rows_in[o] = rows_out[o] = (int32_t)o;
        } else {
            const int32_t* col = nbr + (int64_t)k * m_out;
            for (int64_t o = 0; o < m_out; ++o)
                if (col[o] >= 0) { rows_in.push_back(col[o]); rows_out.push_back((int32_t)o); }
        }
        const int64_t p = (int64_t)rows_in.size();
        if (p == 0) continue;
        buf.resize((size_t)p * c_in);
        res.resize((size_t)p * c_out);
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < p; ++q) memcpy(&buf[(size_t)q * c_in], in + (int64_t)rows_in[q] * c_in, sizeof(float) * c_in);
        const float one = 1.f, zero = 0.f;
        const int n_ = (int)p, cin_ = c_in, cout_ = c_out;
        // row-major res [p, c_out] = buf [p, c_in] @ W [c_in, c_out]  <=>  column-major res^T = W^T buf^T
        sgemm_("N", "N", &cout_, &n_, &cin_, &one, w + (size_t)k * c_in * c_out, &cout_, buf.data(), &cin_, &zero, res.data(),
               &cout_);
#pragma omp parallel for schedule(static)
        for (int64_t q = 0; q < p; ++q) {        // every output row occurs at most once per offset: no conflicts
            float* dst = out + (int64_t)rows_out[q] * c_out;
            const float* src = &res[(size_t)q * c_out];
            for (int c = 0; c < c_out; ++c) dst[c] += src[c];
        }
    }
}

// idx[i] = argmin_j |(b_i s, x, y, z)_full - (b_j s, ...)_part|^2, s = 2 * max over all columns of full, lowest j on ties
void me_ref_argmin_match(const int32_t* full, int64_t m_full, const int32_t* part, int64_t m_part, int64_t* idx) {
    int64_t mx = INT64_MIN;
    for (int64_t i = 0; i < 4 * m_full; ++i) mx = std::max<int64_t>(mx, full[i]);
    const int64_t s = 2 * mx;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m_full; ++i) {
        const int64_t fb = full[4 * i] * s, fx = full[4 * i + 1], fy = full[4 * i + 2], fz = full[4 * i + 3];
        int64_t best = INT64_MAX, bj = 0;
        for (int64_t j = 0; j < m_part; ++j) {
            const int64_t db = fb - part[4 * j] * s, dx = fx - part[4 * j + 1], dy = fy - part[4 * j + 2], dz = fz - part[4 * j + 3];
            const int64_t d = db * db + dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; bj = j; }
        }
        idx[i] = bj;
    }
}

// UNWEIGHTED_AVERAGE: out[v] = mean of feats over inverse == v (fp32 accumulation in point order)
void me_ref_voxel_mean(const float* feats, const int64_t* inverse, int64_t n, int32_t c, int64_t m, float* out) {
    std::vector<float> cnt((size_t)m, 0.f);
    memset(out, 0, sizeof(float) * (size_t)m * c);
    for (int64_t i = 0; i < n; ++i) {
        float* dst = out + inverse[i] * c;
        for (int j = 0; j < c; ++j) dst[j] += feats[i * c + j];
        cnt[(size_t)inverse[i]] += 1.f;
    }
#pragma omp parallel for schedule(static)
    for (int64_t v = 0; v < m; ++v)
        for (int j = 0; j < c; ++j) out[v * c + j] /= cnt[(size_t)v];
}

}  // extern "C"
