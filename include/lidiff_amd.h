/*
 * lidiff_amd.h -- C ABI of the MI355X-native sparse-tensor operator library behind LiDiff.
 *
 * The reference has no FFI of its own: `import MinkowskiEngine as ME` IS its operator
 * interface (lidiff/models/minkunet.py:6, models.py:6, models_refine.py:6,
 * tools/diff_completion_pipeline.py:2).  Every entry point below replaces the native
 * kernel(s) that one of those Python call sites lands in; the call site is cited per
 * function (paths relative to /root/reference/lidiff).  The Python shim
 * lidiff_amd/MinkowskiEngine re-creates the ME symbols on top of this ABI.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing
 *     synchronises, nothing allocates: outputs and workspaces are caller-allocated;
 *   - coordinates are int32 rows (b, x, y, z); every column must lie in [-32768, 32767]
 *     (64-bit packed hash key, 16 bits per column; the single row (32767,32767,32767,32767) is reserved as
 *     the table's empty marker and counts as out of range); violations set bit 0 of *d_status;
 *   - return value 0 = enqueued; != 0 = rejected on the host (bad argument / launch error),
 *     text available from lidiff_last_error() (thread-local);
 *   - row-major, fp32 features, int32 row indices, int64 only where torch indexing wants it.
 */
#ifndef LIDIFF_AMD_H
#define LIDIFF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIDIFF_ABI_VERSION 28
#define LIDIFF_STATUS_KEY_RANGE 1   /* *d_status bit: a coordinate left the 16-bit key range */
#define LIDIFF_STATUS_HASH_FULL 2
                                    /* *d_status bit: hash table too small (cap < 2*rows)     */
#define LIDIFF_STATUS_F16_RANGE 8 /* *d_status bit: a value beyond fp16's range (or Inf / NaN) met the two-piece fp16 split (lidiff_split3_rows / lidiff_spconv_fwd_split3 with pieces = 2) */
#define LIDIFF_STATUS_BOUND 4       /* *d_status bit: a device-side count exceeded the bound the host sized a buffer for (host-read-free
                                       steps, lidiff_tail_map_fill_bounded): nothing was overrun, the results of the step are void */
#define LIDIFF_CONV_SPARSE_MAP 1    /* lidiff_spconv_fwd flags: low-density kernel map (hint) */
#define LIDIFF_CONV_TILE_128 16     /* 64-column layers on 128-row tiles as well (default: 256-row tiles for large maps; A/B measurements) */
#define LIDIFF_CONV_TILE_ONLY 8     /* identity maps (nbr == NULL) through the tile kernel as well, not the row kernel (A/B measurements, tests) */

int lidiff_abi_version(void);
const char* lidiff_last_error(void);

/* Capacity (slots, power of two >= 2*n_rows, >= 1024) of the open-addressing table that
 * lidiff_vox_unique / lidiff_map_stride fill.  Host-only helper. */
int64_t lidiff_hash_capacity(int64_t n_rows);

/* Bytes of scratch lidiff_vox_unique / lidiff_map_stride need for n_rows. Host-only. */
int64_t lidiff_unique_workspace_bytes(int64_t n_rows);

/* TensorField.sparse(), coordinate part -- pipeline:149, models.py:99,202,
 * minkunet.py:135,597 (ME: floor -> CoordinateMapManager::insert_field + hash map).
 * coords_f [n,4] float (integral after LiDiff's round, pipeline:72) -> floor -> int32. */
int lidiff_coords_floor(const float* coords_f, int64_t n_rows, int32_t* coords_i, void* stream);

/* Voxel hashing: unique rows of coords[n,4] in FIRST-OCCURRENCE order.
 *   hkeys[cap] / hvals[cap]: the coordinate map's hash table (filled here; hvals = row id);
 *   uniq[n,4] (first *d_m rows valid), first_idx[n] (point index of each voxel's first
 *   member), inverse[n] (point -> voxel row, int64 as torch indexing wants), *d_m = #voxels.
 * Replaces ME's concurrent hash-map insert + unique_index/inverse_mapping. */
int lidiff_vox_unique(const int32_t* coords, int64_t n_rows,
                      uint64_t* hkeys, int32_t* hvals, int64_t cap,
                      int32_t* uniq, int32_t* first_idx, int64_t* inverse,
                      int32_t* d_m, int32_t* d_status, void* workspace, int32_t preinit, void* stream);
/* preinit (this call, lidiff_map_stride_dev, lidiff_kernel_map_self_dev, lidiff_kernel_map_down_dev, lidiff_vox_mean): != 0 = the
 * caller has ALREADY initialised what the call would otherwise clear itself -- hkeys / hvals / nbr / nbr_down filled with 0xFF bytes,
 * the voxel-mean workspace with zeros -- so that all tables of a coordinate pyramid can live in one pool cleared by ONE memset
 * (round 6: the denoising step queued 90 clears, one per table; lidiff_amd/ops.py build_pyramid*). */

/* UNWEIGHTED_AVERAGE quantisation -- pipeline:77, models.py:171 (ME:
 * MinkowskiSPMMAverageFunction): out[v] = mean of feats[i] over inverse[i]==v.
 * counts[m] (float) is an output too (kept for the backward).  Deterministic: the members of a voxel are summed in
 * 64-bit fixed point with integer atomics (order-free; scale from max |x| and n_rows so that nothing overflows) and the
 * exact sum is rounded to fp32 once.  workspace: lidiff_vox_mean_workspace_bytes(m, c) bytes, 16-byte aligned. */
int64_t lidiff_vox_mean_workspace_bytes(int64_t m, int32_t c);
int lidiff_vox_mean(const float* feats, const int64_t* inverse, int64_t n_rows, int32_t c,
                    int64_t m, float* out, float* counts, void* workspace, int32_t preinit, void* stream);
/* backward of the above: grad_feats[i] = grad_out[inverse[i]] / counts[inverse[i]] */
int lidiff_vox_mean_bwd(const float* grad_out, const int64_t* inverse, const float* counts,
                        int64_t n_rows, int32_t c, float* grad_feats, void* stream);

/* Strided coordinate map -- BasicConvolutionBlock(ks=2,stride=2) minkunet.py:13-29 (ME:
 * CoordinateMapManager::stride): coarse = floor(c / s_out) * s_out on columns 1..3,
 * deduplicated in first-occurrence order over the fine rows.
 *   coarse[n,4] (first *d_m valid), parent[n] = coarse row of each fine row. */
int lidiff_map_stride(const int32_t* coords, int64_t n_rows, int32_t s_out,
                      uint64_t* hkeys, int32_t* hvals, int64_t cap,
                      int32_t* coarse, int32_t* parent,
                      int32_t* d_m, int32_t* d_status, void* workspace, void* stream);

/* The *_dev variants: the same kernels with the input row count read ON THE DEVICE (d_n_rows / d_m: the count the previous
 * level's compaction wrote), n_rows_bound / m_bound being only what the grid, the buffers and the table pitch are sized for
 * (an upper bound: the point count bounds every level).  They let a whole pyramid -- voxelise, the strided maps, the first
 * levels' kernel_size-3 maps and tail-map counts -- be queued without a host read in between; the host then reads all sizes
 * in ONE copy (lidiff_amd.ops.build_pyramid; the reference reads the size of every map as it is made,
 * diff_completion_pipeline.py:162-166 through ME's coordinate manager).  Rows / columns beyond the true count are never
 * written; tables keep the pitch of the bound (callers compact them with a strided copy once the sizes are known). */
int lidiff_map_stride_dev(const int32_t* coords, int64_t n_rows_bound, const int32_t* d_n_rows, int32_t s_out,
                          uint64_t* hkeys, int32_t* hvals, int64_t cap, int32_t* coarse, int32_t* parent,
                          int32_t* d_m, int32_t* d_status, void* workspace, int32_t preinit, void* stream);

/* Kernel map (rulebook) as a neighbour table -- MinkowskiConvolution ks=3 (minkunet.py:
 * 53-66,94,97,156,159,512,515) and ks=2/stride 2 (13-29) (ME: CoordinateMapManager::
 * kernel_map).  nbr[k*m_out + o] = row of the INPUT map holding out_coords[o] +
 * offset_k*step, or -1.  Offsets iterate x fastest; ks odd: centred {-1,0,1}; ks even:
 * {0,1}.  (hkeys,hvals,cap) is the INPUT map's table. */
int lidiff_kernel_map(const int32_t* out_coords, int64_t m_out,
                      const uint64_t* hkeys_in, const int32_t* hvals_in, int64_t cap_in,
                      int32_t ks, int32_t step, int32_t* nbr, void* stream);

/* lidiff_kernel_map for kernel_size 3 of a coordinate map onto ITSELF (out_coords = the rows of the table's map, the
 * stride-1 convolutions of minkunet.py:53-66,94,97): the same table bit for bit, from half the lookups -- only the 13
 * offsets below the centre are probed, every hit also fills its mirror entry nbr[26 - k][hit] = row, the centre is the
 * identity.  Fills nbr [27, m] itself (no pre-initialisation needed). */
int lidiff_kernel_map_self(const int32_t* coords, int64_t m, const uint64_t* hkeys, const int32_t* hvals, int64_t cap,
                           int32_t step, int32_t* nbr, void* stream);
/* ... with the row count on the device: nbr [27, m_bound] (pitch m_bound), columns >= *d_m stay -1. */
int lidiff_kernel_map_self_dev(const int32_t* coords, int64_t m_bound, const int32_t* d_m, const uint64_t* hkeys,
                               const int32_t* hvals, int64_t cap, int32_t step, int32_t* nbr, int32_t preinit, void* stream);

/* The kernel_size-2 / stride-2 map of a strided convolution (fine map -> its coarse map, minkunet.py:13-29) from the
 * parent array lidiff_map_stride returned: nbr_down [8, m_coarse] -- the table lidiff_kernel_map(coarse coords, fine
 * table, ks 2, step ts_fine) builds, bit for bit, from one pass over the fine rows instead of 8 lookups per coarse row. */
int lidiff_kernel_map_down(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine, int32_t ts_fine,
                           int64_t m_coarse, int32_t* nbr_down, void* stream);
/* ... with the fine map's row count on the device (*d_m_fine <= m_fine_bound); nbr_down [8, m_coarse_bound] (pitch = the bound of
 * the coarse map), columns without a fine row stay -1. */
int lidiff_kernel_map_down_dev(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine_bound, const int32_t* d_m_fine,
                               int32_t ts_fine, int64_t m_coarse_bound, int32_t* nbr_down, int32_t preinit, void* stream);

/* Kernel map of MinkowskiConvolutionTranspose(ks=2,stride=2) -- minkunet.py:32-46 (ME:
 * swapped fine->coarse map): nbr_up[k*m_fine + j] = parent[j] if k == kernel index of
 * (fine_coords[j] - coarse coordinate)/ts_fine (x fastest) else -1. */
int lidiff_kernel_map_up(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine,
                         int32_t ts_fine, int32_t* nbr_up, void* stream);
/* ... with the row count on the device: nbr_up [8, m_fine_bound], columns >= *d_m_fine are -1. */
int lidiff_kernel_map_up_dev(const int32_t* fine_coords, const int32_t* parent, int64_t m_fine_bound, const int32_t* d_m_fine,
                             int32_t ts_fine, int32_t* nbr_up, void* stream);

/* Morton (Z-order) key of every row of a coordinate map at tensor stride ts (batch index above 3 x 16
 * interleaved bits of x, y, z / ts).  No reference counterpart: ME processes rows in hash order; here the
 * argsort of these keys is handed to lidiff_spconv_fwd as `row_order`, so that a 128-row tile is a compact
 * block of space and its gathers hit the L2 (results are independent of the order). */
int lidiff_morton_keys(const int32_t* coords, int64_t m, int32_t ts, int64_t* keys, void* stream);

/* ME-layout rulebook from a neighbour table: for every k the (in,out) pairs sorted by
 * out row, concatenated; offset_ptr[K+1] (device).  Two-phase: call with pairs_in == NULL
 * to fill offset_ptr only (count pass), then with arrays of offset_ptr[K] entries. */
int lidiff_rulebook_compact(const int32_t* nbr, int32_t k_vol, int64_t m_out,
                            int32_t* offset_ptr, int32_t* pairs_in, int32_t* pairs_out,
                            void* workspace, void* stream);
int64_t lidiff_rulebook_workspace_bytes(int32_t k_vol, int64_t m_out);

/* Tail map of a kernel_size-3 / stride-1 kernel map, for the two-pass convolution of low-density maps (see the `tail`
 * arguments of lidiff_spconv_fwd): the pairs of every offset but `skip` (13, the centre: the identity on such a map) as P
 * rows sorted by (offset, output row).  Two phases on one workspace (lidiff_tail_map_workspace_bytes):
 *   phase 1 (tail_nbr == NULL): offset_ptr[k_vol + 1] (pairs before each offset; [k_vol] = P) and row_ptr[m_out + 1]
 *     (CSR over the map's output rows) -- the host reads P = offset_ptr[k_vol];
 *   phase 2: tail_nbr [k_vol, P] (row p: the pair's input row at its own offset, -1 elsewhere: a kernel map whose output
 *     rows are the pairs) and idx [P] (the pairs of output row o, ascending offset: idx[row_ptr[o] .. row_ptr[o+1])).
 * No reference counterpart: ME walks its per-offset in/out lists once per convolution (minkunet.py:53-66). */
int64_t lidiff_tail_map_workspace_bytes(int32_t k_vol, int64_t m_out);
int lidiff_tail_map(const int32_t* nbr, int32_t k_vol, int64_t m_out, int32_t skip, int32_t* offset_ptr,
                    int32_t* row_ptr, int64_t n_pairs, int32_t* tail_nbr, int32_t* idx, void* workspace, void* stream);
/* ... over a table of pitch m_bound whose row count lives on the device (both phases must use the same m_bound and
 * workspace): row_ptr [m_bound + 1], of which [0, *d_m] is the CSR. */
int lidiff_tail_map_dev(const int32_t* nbr, int32_t k_vol, int64_t m_bound, const int32_t* d_m, int32_t skip,
                        int32_t* offset_ptr, int32_t* row_ptr, int64_t n_pairs, int32_t* tail_nbr, int32_t* idx,
                        void* workspace, void* stream);
/* Phase 2 of lidiff_tail_map_dev when the PAIR COUNT stays on the device as well (a denoising step without a host read): the
 * pair list form of the tail map -- pair_in [n_pairs_bound] (the input row of pair p; what lidiff_spconv_fwd_pairs walks with
 * offset_ptr) and idx [n_pairs_bound] -- into buffers sized by a bound.  If offset_ptr[k_vol] > n_pairs_bound the pairs behind
 * the bound are dropped and LIDIFF_STATUS_BOUND is raised in *d_status (the consumers clamp their ranges to the bound, so
 * nothing is overrun; the caller redoes the step with exact sizes). */
int lidiff_tail_map_fill_bounded(const int32_t* nbr, int32_t k_vol, int64_t m_bound, const int32_t* d_m, int32_t skip,
                                 const int32_t* offset_ptr, const int32_t* row_ptr, int64_t n_pairs_bound, int32_t* pair_in,
                                 int32_t* idx, int32_t* d_status, void* workspace, void* stream);

/* n_words device int32 words (the row counts of a coordinate pyramid) and *d_status (nullable) written into host-visible memory
 * (host_mapped: pinned, device-mapped; [0] = seq, [1] = status, [2 ..] = the words), the sequence number last behind a
 * system-scope fence.  The host side of a denoising loop learns the sizes of step i from it while step i + 1 is already queued
 * -- no device->host copy, no synchronisation (DiffCompletion, pipeline:155-169).  n_words <= 62. */
/* The device-side address of pinned host memory (hipHostGetDevicePointer), for host_mapped below; non-zero status when the
 * memory is not mapped into the device's address space (the caller then copies instead).  Host-only helper. */
int lidiff_host_device_pointer(void* host_ptr, void** dev_ptr);
int lidiff_publish_words(const int32_t* words, int32_t n_words, const int32_t* d_status, int32_t* host_mapped, int32_t seq,
                         void* stream);
/* Weight layout of the sparse convolution.  MinkowskiConvolution.kernel is [K, c_in, c_out] row-major
 * (minkunet.py:17,36,53,61; [c_in, c_out] for kernel_size 1, :72).  The HIP kernel consumes it in MFMA
 * fragment order: [K][slab = ceil(c_in/32)][c_out/16][j 0..1][lane 0..63][e 0..3] with
 * k_in = 32*slab + 16*j + 4*(lane>>4) + e (zero beyond c_in) and col = 16*nt + (lane&15), so that every
 * wave reads its [32 x 16] piece of W[k] as one coalesced 2 KB run straight into VGPRs.  Pack once per
 * weight update (the Python shim caches the packed copy per Parameter version). c_out % 16 == 0. */
int64_t lidiff_spconv_packed_weight_floats(int32_t k_vol, int32_t c_in, int32_t c_out);
int lidiff_spconv_pack_weights(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out,
                               float* w_packed, void* stream);

/* Sparse convolution forward -- MinkowskiConvolution / MinkowskiConvolutionTranspose
 * (ME: ConvolutionForwardGPU gather-GEMM-scatter), output-stationary, fp32 MFMA:
 *   out[o, :] = epilogue( sum_k  in[nbr[k,o], :] @ w[k] ),  in = [in_a | in_b] column-wise
 *   (in_b may be NULL; it fuses ME.cat(in_a, in_b), minkunet.py:464,474,484,494).
 *   nbr == NULL means K == 1 identity map (kernel_size=1, minkunet.py:72).
 *   epilogue: v = acc*ep_scale[c] + ep_shift[c] (either may be NULL) ; v += residual[o,c]
 *   (may be NULL) ; relu if relu != 0.   (eval-mode MinkowskiBatchNorm + MinkowskiReLU +
 *   ResidualBlock add, minkunet.py:23-24,59-60,79.)
 * w_packed: lidiff_spconv_pack_weights of the [K, c_in_a + c_in_b, c_out] kernel.
 * row_order: NULL, or a permutation of the output rows: tile t covers output rows row_order[128 t ..] and
 *   column j of `nbr` then belongs to output row row_order[j] (the caller permutes the table's columns).
 * replicas: R >= 1 feature matrices stacked row-wise ([R*m_in, c] in, [R*m_out, c_out] out / residual) share
 *   the kernel map and the weights -- the conditional / unconditional pair of classifier-free guidance
 *   (pipeline:148-153) in one launch; m_in / m_out are per replica.
 * flags: LIDIFF_CONV_SPARSE_MAP = the kernel map is expected to hold only a few pairs per offset and
 *   128-row tile (a performance hint, results are identical): such tiles pack several offsets into
 *   one 128-row stage.
 *   Identity maps (nbr == NULL, no row order; channel widths multiples of 16 with c_in in {32, 64, 96, 128, 192}, c_out a
 *   multiple of 32) run as a streaming row GEMM (spconv_rows.hip: W column tile resident in LDS, rows straight from HBM
 *   into the MFMA operands, no barrier, 16-byte stores) with bit-identical results; LIDIFF_CONV_TILE_ONLY keeps them on
 *   the tile kernel.  lidiff_spconv_fwd_kernel_id answers which kernel a call will take.
 * tail / tail_ptr / tail_idx (all null, or all set): rows added to the convolution sum BEFORE the epilogue through a
 *   CSR over the output rows: out[o] += sum over q in [tail_ptr[o], tail_ptr[o+1]) of tail[tail_idx[q], :]
 *   (tail [replicas * tail_rows, c_out]).  This is how a low-density kernel_size-3 map (stride-1 / 2 levels of a noisy
 *   scan: ~1.1-1.5 neighbours per voxel) is convolved: the centre offset is the identity map -- one dense
 *   [M, c_in] x [c_in, c_out] pass over contiguous rows, launched with nbr == NULL and W[13] -- and the few pairs of
 *   the other 26 offsets are multiplied beforehand, grouped by offset (weight stationary: W[k] is read once per 128
 *   pairs instead of once per output tile that has a single pair of offset k), into `tail`, one row per pair
 *   (lidiff_amd/MinkowskiEngine CoordinateManager.tail_map).  Fixed summation order: deterministic.
 * d_m_out (nullable): the number of valid output rows lives on the device (*d_m_out <= m_out); m_out is then only its BOUND --
 *   it still sizes the grid, the row pitch of nbr and the replica pitch of in / out / residual -- and tiles behind the count
 *   leave at once.  A denoising step whose map sizes never reach the host (DiffCompletion). */
int lidiff_spconv_fwd(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                      const float* w_packed, const int32_t* nbr, int32_t k_vol,
                      int64_t m_in, int64_t m_out, int32_t c_out, float* out,
                      const float* ep_scale, const float* ep_shift, const float* residual,
                      int32_t relu, const int32_t* row_order, int32_t replicas, int32_t flags,
                      const float* tail, const int32_t* tail_ptr, const int32_t* tail_idx, int64_t tail_rows,
                      const int32_t* d_m_out, void* stream);

/* Which kernel lidiff_spconv_fwd runs for these arguments: 0 = tile kernel (spconv.hip),
 * 2 = row kernel (spconv_rows.hip), 3 = thin-input kernel (c_in <= 4, c_out == 32: the stems; spconv_rows.hip; equal to the
 * tile kernel up to fp32 summation order).  has_nbr / has_row_order: whether those pointers are non-null.  For profilers and tests. */
int32_t lidiff_spconv_fwd_kernel_id(int32_t c_in_a, int32_t c_in_b, int32_t c_out, int32_t k_vol, int32_t has_nbr,
                                    int32_t has_row_order, int32_t flags);

/* Sparse convolution forward over a map in which every output row has exactly ONE pair, given as a pair list grouped by
 * kernel offset (ME-layout rulebook, lidiff_rulebook_compact): for p in [offset_ptr[k], offset_ptr[k+1])
 *     out[pair_out[p], :] = epilogue( [in_a | in_b][pair_in[p], :] @ w[k] )          (pair_out == NULL: out row p)
 * -- MinkowskiConvolutionTranspose with kernel_size 2 / stride 2 (minkunet.py:36: every fine voxel has one parent) and the
 * tail pass of the centre + tail scheme above (one output row per pair, pair_out == NULL, m_out == n_pairs).  A streaming
 * row GEMM (spconv_rows.hip): the offset's W column tile resident in LDS, gathered rows straight from HBM into the MFMA
 * operands, 16-byte stores; results bit-identical to lidiff_spconv_fwd over the same map.  Input widths multiples of 16
 * summing to 32 / 64 / 96 / 128, c_out a multiple of 32 (lidiff_spconv_fwd_pairs_supported answers 1 / 0); epilogue and
 * replicas as lidiff_spconv_fwd (m_in / m_out per replica; the pair list is shared). */
int32_t lidiff_spconv_fwd_pairs_supported(int32_t c_in_a, int32_t c_in_b, int32_t c_out);
int lidiff_spconv_fwd_pairs(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b, const float* w_packed,
                            int32_t k_vol, const int32_t* pair_in, const int32_t* pair_out, const int32_t* offset_ptr,
                            int64_t n_pairs, int64_t m_in, int64_t m_out, int32_t c_out, float* out,
                            const float* ep_scale, const float* ep_shift, const float* residual, int32_t relu,
                            int32_t replicas, void* stream);

/* lidiff_spconv_fwd with bf16 matrix operands and fp32 accumulation (v_mfma_f32_16x16x32_bf16): the mixed-precision TRAINING
 * convolution (train.py under bf16; models.py:180-217) -- forward and, over the swapped map with W^T, the input gradient.
 * Features, BatchNorm and the epilogue stay fp32 in HBM; each gathered input value is rounded to bf16 (nearest even) as it enters
 * the matrix unit, the weights when they are packed (lidiff_spconv_pack_weights_bf16 with planes = 1, once per weight version).
 * Equal to lidiff_spconv_fwd on bf16-rounded inputs and weights up to fp32 summation order.  `planes` must be 1 (rounds 1-5
 * also built fp32-accurate results from 2 / 3 pieces per operand on this kernel; that is lidiff_spconv_fwd_split3 now).  The
 * weight packer cuts 1..3 pieces: piece 0 = bf16(w), piece 1 = bf16(w - piece 0), piece 2 = bf16(w - piece 0 - piece 1).
 * Input widths and c_out multiples of 32; same map / epilogue / replica arguments as lidiff_spconv_fwd (no row order, no
 * tail).
 * in_bf16 != 0 (1 = the library picks the kernel, 2 / 3 force the ring / the two-stage kernel -- bit-identical,
 * for A/B measurements and tests): in_a / in_b point at bf16 rows -- the shadow copy of the fp32 feature matrix that its producer or
 * lidiff_cast_bf16 left (bf16 activations in HBM, the bf16 training configuration): half the gather traffic and requests, no
 * conversion in the kernel; the same operands, products and order of sums, i.e. bit-identical to in_bf16 = 0 on the fp32 rows.
 * lidiff_cast_bf16: n fp32 values -> n bf16 values, round to nearest even. */
int lidiff_cast_bf16(const float* src, int64_t n, void* dst, void* stream);
int64_t lidiff_spconv_packed_weight_bf16_elems(int32_t k_vol, int32_t c_in, int32_t c_out, int32_t planes);
int lidiff_spconv_pack_weights_bf16(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out, int32_t planes,
                                    void* w_packed, void* stream);
int lidiff_spconv_fwd_bf16(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b, const void* w_packed,
                           int32_t planes, const int32_t* nbr, int32_t k_vol, int64_t m_in, int64_t m_out, int32_t c_out,
                           float* out, const float* ep_scale, const float* ep_shift, const float* residual, int32_t relu,
                           int32_t replicas, int32_t in_bf16, void* stream);

/* lidiff_spconv_fwd for the dense levels with the contraction on the bf16 matrix pipe from THREE-WAY SPLIT operands (round 6;
 * MinkowskiConvolution, minkunet.py:53-66,184-259, eval-mode fused plan): fp32 in, fp32 out, fp32 accuracy.  Every fp32 value is
 * the exact sum of three bf16 pieces x0 + x1 + x2 (round to nearest even at each cut); the six products x_i w_j with i + j <= 2 --
 * each exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16 -- leave out only terms below 2^-24 |x w|: the error against
 * float64 equals the fp32 MFMA's own (tests/test_gpu_kernels.py::test_spconv_split3_*; profiles/r06_split3.txt).
 *   lidiff_split3_rows: fp32 [m][c] -> bf16 [m][3][c] (pieces = 3; the operand layout; c % 8 == 0);
 *   weights: lidiff_spconv_pack_weights_bf16 with planes = 3;
 *   lidiff_spconv_fwd_split3: in_a3 / in_b3 = split matrices of `replicas` stacked feature matrices (fused ME.cat as in
 *   lidiff_spconv_fwd), nbr / k_vol / m_in / m_out / epilogue / replicas / d_m_out / row_order as there; out_planes (nullable): the
 *   output ALSO as bf16 [replicas * m_out][3][c_out] -- the next dense convolution's operand, cut in the epilogue.
 *   Shapes: c_in_a, c_in_b multiples of 32, c_out a multiple of 64 (lidiff_spconv_fwd_split3_supported). */
int lidiff_split3_rows(const float* src, int64_t m, int32_t c, int32_t pieces, void* dst, int32_t* d_status, const int32_t* d_rows,
                       int64_t pitch, void* stream);   /* d_rows / pitch (pieces = 2, nullable): replicas of `pitch` rows with *d_rows valid ones each -- rows behind them are not cut (uninitialised memory must not raise the range flag) */
int32_t lidiff_spconv_fwd_split3_supported(int32_t c_in_a, int32_t c_in_b, int32_t c_out);
int lidiff_spconv_fwd_split3(const void* in_a3, int32_t c_in_a, const void* in_b3, int32_t c_in_b, const void* w_packed3,
                             const int32_t* nbr, int32_t k_vol, int64_t m_in, int64_t m_out, int32_t c_out, float* out,
                             void* out_planes, const float* ep_scale, const float* ep_shift, const float* residual,
                             int32_t relu, int32_t replicas, const int32_t* d_m_out, const int32_t* row_order, int32_t pieces,
                             float out_scale, int32_t* d_status, void* stream);
/* pieces = 2 (opt-in; pieces = 3 is everything described above): the same kernel on TWO fp16 pieces per operand and three
 * products (x0 w1, x1 w0, x0 w0) -- half the matrix work, 4 instead of 6 bytes per gathered element.  The operands then carry 22
 * bits (fp32: 24): a reduction of precision, small beside the rounding of the fp32 sums (profiles/r06_f16x2.txt), but a
 * reduction; fp16's range applies to the features (|x| <= 65504: a value beyond it, Inf or NaN raises LIDIFF_STATUS_F16_RANGE in
 * *d_status -- nullable -- and the results are void).  Layouts: fp16 [m][2][c] rows (lidiff_split3_rows, or cut by the kernel's
 * epilogue into out_planes), weights packed by lidiff_spconv_pack_weights_f16x2 AFTER multiplication by `scale`, a power of two the
 * caller picks so that the second pieces are normal fp16 numbers (max |w| * scale in [2^11, 2^12) works); out_scale = 1 / scale
 * undoes it in the epilogue (exact). */
int lidiff_spconv_pack_weights_f16x2(const float* w, int32_t k_vol, int32_t c_in, int32_t c_out, float scale, void* w_packed,
                                     int32_t* d_status, void* stream);
/* Rows sorted by their neighbour sets: lidiff_row_mask_keys writes, per row of a table nbr [k_vol][m], one bit per offset = "has
 * a neighbour under it" (k_vol = 27: the offsets present least often on a scan's surfaces -- out of the horizontal plane, corners
 * before edges before faces -- in the leading bits; otherwise bit k for offset k; the centre offset also as bit 27: a descending
 * sort puts the valid rows of a table handed over at its bound first).  The caller sorts (stable, descending), permutes the table's columns (nbr[:, order]) and passes `row_order` = order
 * (tile row -> output row) to lidiff_spconv_fwd_split3: results are the same values, but whole 16-row blocks of a tile then lack
 * an offset and are neither gathered nor multiplied (the kernel's block masks) -- 30 % fewer MFMAs at stride 8, 70 % at stride 4. */
int lidiff_row_mask_keys(const int32_t* nbr, int32_t k_vol, int64_t m, int32_t* keys, void* stream);

/* Weight gradient of lidiff_spconv_fwd (training path, models.py:180-217; ME: ConvolutionBackwardGPU):
 * dw[k] += gather(in)[pairs_in of offset k]^T @ grad_out[pairs_out of offset k], in = [in_a | in_b], over the
 * ME-layout rulebook of the kernel map (lidiff_rulebook_compact: pairs sorted by offset, offset_ptr [K+1] on the
 * device, n_pairs = offset_ptr[K] known to the host); all three null = identity map (K = 1).  The pairs of an offset are
 * cut into slices, one workgroup each.  With `workspace` (lidiff_spconv_bwd_w_workspace_floats floats) every slice stores
 * its partial tile and a second kernel sums them in slice order: dw is DETERMINISTIC and need not be zeroed.  With
 * workspace == NULL the slices meet in dw by fp32 atomics (as ME's GPU path does): dw must be zeroed by the caller and is
 * reproducible only up to the order of those adds.  Channel counts multiples of 4.  The input gradient needs no entry
 * point of its own: it is lidiff_spconv_fwd over the swapped map with W^T. */
int64_t lidiff_spconv_bwd_w_workspace_floats(int32_t c_in, int32_t c_out, int32_t k_vol, int64_t n_pairs);
int lidiff_spconv_bwd_w(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b,
                        const float* grad_out, const int32_t* pairs_in, const int32_t* pairs_out,
                        const int32_t* offset_ptr, int64_t n_pairs, int32_t k_vol,
                        int64_t m_in, int64_t m_out, int32_t c_out, float* dw, float* workspace, void* stream);
/* lidiff_spconv_bwd_w with bf16 operands (bf16 training, planes = 1 of lidiff_spconv_fwd_bf16): the gathered rows of `in`
 * and `grad_out` are rounded to bf16 (nearest even) on their way into v_mfma_f32_16x16x32_bf16, sums in fp32.  Same
 * arguments, tiles, pair slices and workspace (lidiff_spconv_bwd_w_workspace_floats) as lidiff_spconv_bwd_w.
 * in_bf16: in_a, in_b AND grad_out point at bf16 rows (see lidiff_spconv_fwd_bf16): bit-identical sums, 8-byte loads. */
int lidiff_spconv_bwd_w_bf16(const float* in_a, int32_t c_in_a, const float* in_b, int32_t c_in_b, const float* grad_out,
                             const int32_t* pairs_in, const int32_t* pairs_out, const int32_t* offset_ptr, int64_t n_pairs,
                             int32_t k_vol, int64_t m_in, int64_t m_out, int32_t c_out, float* dw, float* workspace,
                             int32_t in_bf16, void* stream);

/* The network's last lines as one launch (head.hip): out[r n + i, :] = W2 leaky(W1 feats[r m_rows + inverse[i], :] + b1) + b2 --
 * SparseTensor.slice(field).F followed by `self.last` = Linear(C, hidden), LeakyReLU(slope), Linear(hidden, c_out)
 * (minkunet.py:390,497).  feats: `replicas` stacked voxel matrices of m_rows rows each (the CFG pair shares the map, hence
 * `inverse` [n_points]); w1_t: the first Linear's weight TRANSPOSED, [C][hidden]; w2 [c_out][hidden] as torch stores it.
 * Sums over the channels in ascending order: the same bits for any m_rows.  lidiff_slice_head_supported: the shapes it takes
 * (C % 4 == 0, hidden 20, c_out 3 -- LiDiff's head); other heads stay on the caller's GEMMs. */
int32_t lidiff_slice_head_supported(int32_t c, int32_t hidden, int32_t c_out);
int lidiff_slice_head(const float* feats, const int64_t* inverse, int64_t n_points, int64_t m_rows, int32_t replicas, int32_t c,
                      const float* w1_t, const float* b1, int32_t hidden, const float* w2, const float* b2, int32_t c_out,
                      float slope, float* out, void* stream);

/* Row gather / scatter-add -- SparseTensor.slice(field).F minkunet.py:497,619 and the
 * x_part.F[idx] of match_part_to_full minkunet.py:418; scatter-add is their backward. */
int lidiff_gather_rows(const float* src, const int64_t* idx, int64_t n_rows, int32_t c,
                       float* dst, void* stream);
/* The same sum without atomics: dst[o, :] = sum of src[order[q], :] over q in [ptr[o], ptr[o + 1]), in list order -- `order` the
 * sources sorted by destination row (stable: source order inside a destination), `ptr` [m + 1] the CSR over the m destination
 * rows.  Deterministic: the backward of SparseTensor.slice and of the conditioning gathers in training (models.py:180-217).
 * workspace (nullable; lidiff_segment_sum_workspace_bytes(n_sources, c)): destinations with more than 64 sources -- the
 * unconditional branch of a training step, models.py:192-195, gathers ~180 000 rows from each of 2 part voxels -- are summed in
 * chunks of 1024 sources by one workgroup per (chunk, 32 channels), the chunk sums then in chunk order: a fixed order as well;
 * without it every segment takes the one-thread path.  n_sources = rows of src / entries of order. */
int64_t lidiff_segment_sum_workspace_bytes(int64_t n_sources, int32_t c);
int lidiff_segment_sum_rows(const float* src, const int64_t* order, const int64_t* ptr, int64_t m, int32_t c, float* dst,
                            int64_t n_sources, void* workspace, void* stream);

/* The conditioning multiply x * w, w = latemp(cat(latent(match), temp)) -- minkunet.py:424-431 etc.: for one batch
 * (one time embedding) every layer of that MLP is row-wise and commutes with the gather of match_part_to_full, so w
 * is table[idx] with table = the MLP evaluated on the few part rows: dst[r,:] = x[r,:] * table[idx[r],:] in one pass.
 * c % 4 == 0, pointers 16-byte aligned.  idx == NULL: every row takes table row 0 (the one-voxel unconditional branch: a
 * broadcast).  d_n_rows (nullable): the number of valid rows lives on the device, n_rows is its bound. */
int lidiff_gather_mul_rows(const float* x, const float* table, const int64_t* idx, int64_t n_rows, int32_t c,
                           float* dst, const int32_t* d_n_rows, void* stream);

/* Hidden layer of the conditioning MLPs -- minkunet.py:424-431 (latemp_* applied to cat(latent(match), temp)):
 * with the row-wise Linear commuted in front of the gather, dst[r,:] = leaky_relu(src[idx[r],:] + bias[:], slope)
 * in one pass (src = first-Linear output on the few part rows, bias = the time-embedding half + Linear bias).
 * c % 4 == 0, pointers 16-byte aligned. */
int lidiff_gather_bias_leaky(const float* src, const int64_t* idx, const float* bias, int64_t n_rows, int32_t c,
                             float slope, float* dst, void* stream);

/* MinkUNetDiff.match_part_to_full -- minkunet.py:403-418 (pykeops argKmin(1)): for every
 * full row the index of the nearest part row by squared L2 over (b*scale, x, y, z), ties to
 * the lowest index. scale = 2 * (*d_max_coord) as in the reference (d_max_coord: device int32,
 * the max over ALL columns of full).
 * d_gate (nullable; one device int32 of the caller's): batches of several scans (training, models.py:180-217) -- every row is
 * first matched against the part rows of its OWN batch element only (part rows grouped by ascending batch index, as a
 * voxelised batch is); a winner closer than scale^2, the batch term alone of every other element's rows, is final.  A check
 * kernel sets *d_gate when some row's winner is not (or the part rows are not grouped) and the unrestricted search then runs over
 * the same idx[]: the same indices as with d_gate == NULL in every case, B-fold less work in the usual one. */
int lidiff_nn_match(const int32_t* full, int64_t m_full, const int32_t* part, int64_t m_part,
                    const int32_t* d_max_coord, int64_t* idx, int32_t* d_gate, void* stream);
/* The same search for a map whose row count is still on the device (a coordinate pyramid before its one host read,
 * lidiff_map_stride_dev): full has room for m_full_bound rows, the first *d_m_full are valid; d_max_coord [1] is computed here
 * (max over all columns of the valid rows) and idx [m_full_bound] written for the valid rows only.  Lets the five part -> full
 * matches of a step run beside the pyramid's own kernels instead of behind its host read. */
int lidiff_nn_match_dev(const int32_t* full, int64_t m_full_bound, const int32_t* d_m_full, const int32_t* part, int64_t m_part,
                        int32_t* d_max_coord, int64_t* idx, void* stream);

/* The generic brute-force arg-min behind the reference's pykeops expression (minkunet.py:412-416:
 * ((LazyTensor(f[:,None,:]) - LazyTensor(p[None,:,:]))**2).sum(-1).argKmin(1, dim=1)): rows are float4
 * (dimension <= 4, zero padded by the caller), idx[i] = the lowest j minimising |a_i - b_j|^2 in fp32 with the
 * summation order of the scalar expression.  Binding target of lidiff_amd/compat's LazyTensor shim, which lets the
 * reference's own minkunet.py run unmodified on this library. */
int lidiff_argmin_rows_f32(const float* a, int64_t n, const float* b, int64_t m, int64_t* idx, void* stream);

/* Training-mode batch normalisation over the rows of a feature matrix [m, c] -- MinkowskiBatchNorm = nn.BatchNorm1d on F
 * (minkunet.py:23,59,79; the training step of models.py:180-217).  One pass over [m, c] per kernel, double accumulators,
 * partial sums combined in a fixed order: deterministic.  c a multiple of 4, pointers 16-byte aligned.
 *   lidiff_bn_stats : mean[c], var[c] (BIASED: the normaliser), invstd[c] = 1 / sqrt(var + eps); running_mean / running_var
 *                     (nullable) are updated in the same launch as nn.BatchNorm1d updates them: r = (1 - momentum) r + momentum x
 *                     with the UNBIASED variance var * m / (m - 1).  workspace: lidiff_bn_workspace_bytes(c).
 *   lidiff_bn_apply : y = (x - mean) * invstd * gamma + beta (gamma / beta nullable) [+ residual, nullable: the ResidualBlock's
 *                     shortcut, minkunet.py:79], ReLU on request (relu != 0).
 *   lidiff_bn_bwd   : sum_dy[c] = sum dy, sum_dy_xmu[c] = sum dy * (x - mean) (d beta and, times invstd, d gamma), and -- unless
 *                     dx == NULL -- dx = (dy - sum_dy / m - (x - mean) * invstd^2 * sum_dy_xmu / m) * invstd * gamma.
 *                     y_relu != NULL: the forward applied ReLU and y_relu is its output -- dy counts only where y_relu > 0.
 *                     d_residual != NULL: receives that (masked) dy, the gradient of the forward's residual operand.
 *   y_bf16 / dx_bf16 (nullable): a bf16 copy of y / dx (round to nearest even) written by the same launch -- the shadow rows
 *                     the bf16 convolutions gather (lidiff_spconv_fwd_bf16 in_bf16), instead of a lidiff_cast_bf16 pass. */
int64_t lidiff_bn_workspace_bytes(int32_t c);
int lidiff_bn_stats(const float* x, int64_t m, int32_t c, float eps, float* mean, float* var, float* invstd,
                    float* running_mean, float* running_var, float momentum, void* workspace, void* stream);
int lidiff_bn_apply(const float* x, int64_t m, int32_t c, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, const float* residual, int32_t relu, float* y, void* y_bf16, void* stream);
int lidiff_bn_bwd(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                  const float* invstd, const float* gamma, float* sum_dy, float* sum_dy_xmu, float* dx, float* d_residual,
                  void* workspace, void* dx_bf16, void* stream);

/* The same normalisation with the statistics shared over a process group -- ME.MinkowskiSyncBatchNorm, which
 * train.py:90 / train_refine.py:58 (convert_sync_batchnorm) put in place of every MinkowskiBatchNorm under DDP.  The library never
 * communicates: it hands the caller the LOCAL per-channel sums as fp64, the caller all-reduces them (RCCL; SUM is exact enough in
 * fp64 to make every rank derive bit-identical statistics) and hands them back.
 *   (Every piece below takes m = 0 -- and x / dy / y NULL then: a rank without a row of some layer still joins the layer's
 *    collectives, with zero sums and a count of zero, as torch's SyncBatchNorm lets it.)
 *   lidiff_bn_sums            : sums[0..c) = sum x, sums[c..2c) = sum x^2 over this rank's m rows, sums[2c] = m.  The caller
 *                               all-reduces all 2c + 1 doubles in ONE collective.
 *   lidiff_bn_stats_from_sums : mean / biased var / invstd of lidiff_bn_stats from such (all-reduced) sums, count = sums[2c]
 *                               read on the device (no host round trip); running estimates updated as lidiff_bn_stats does,
 *                               with the GLOBAL count.  lidiff_bn_apply then runs unchanged.
 *   lidiff_bn_bwd_sums        : sums[0..c) = sum dy, sums[c..2c) = sum dy * (x - mean) over this rank's rows (mean = the GLOBAL
 *                               mean; dy masked by y_relu > 0 as in lidiff_bn_bwd): d beta and, times invstd, d gamma of this
 *                               rank -- and, all-reduced, the two projections dx needs.
 *   lidiff_bn_bwd_apply       : dx (and d_residual) of lidiff_bn_bwd from the all-reduced sums and the device-resident global row
 *                               count; sum_dy / sum_dy_xmu receive the fp32 copies the kernel reads. */
int lidiff_bn_sums(const float* x, int64_t m, int32_t c, double* sums, void* workspace, void* stream);
int lidiff_bn_stats_from_sums(const double* sums, int32_t c, float eps, float* mean, float* var, float* invstd,
                              float* running_mean, float* running_var, float momentum, void* stream);
int lidiff_bn_bwd_sums(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                       double* sums, void* workspace, void* stream);
int lidiff_bn_bwd_apply(const float* dy, const float* x, const float* y_relu, int64_t m, int32_t c, const float* mean,
                        const float* invstd, const float* gamma, const double* sums, const double* count, float* sum_dy,
                        float* sum_dy_xmu, float* dx, float* d_residual, void* dx_bf16, void* stream);

/* The boundary between two denoising steps as one launch -- DiffCompletion.completion_loop, pipeline:148-153 (classifier-free
 * guidance), :161-163 (offsets, dpm_scheduler.step: sde-dpmsolver++) and :164 with :68-84 (points_to_tensor of the
 * new points): per point and axis, every operation rounded on its own in the precision torch-on-the-GPU gives it,
 *   eps  = e_u + w (e_c - e_u)                         (fp32)
 *   off  = (double) x_t - x_init ;  x0 = (off - (double)(sigma_t eps)) * inv_alpha_t                     (fp64; sigma_t eps in fp32)
 *   prev = c_sample off + c_m0 x0 [+ c_d1 (inv_r0 (x0 - m_prev))] [+ c_noise noise]                      (fp64, left to right)
 *   feats = (float)(x_init + prev) ;  coords = (b', rint(feats * inv_resolution)) as int32
 * m_prev NULL: first-order update; noise NULL: no noise term.  b' = the batch index i / n_per_batch, passed through
 * rint(b * inv_resolution) when scale_batch_column != 0 (pipeline:72 divides the batch column too).
 * x0_out [n,3] fp64 (the solver's next m_prev), feats_out [n,3] fp32, coords_out [n,4] int32: what
 * TensorField(features, coordinates) of the next step takes.  The coefficients are host doubles (the scheduler's tables). */
int lidiff_cfg_dpm_step(const float* eps_cond, const float* eps_uncond, float w, const float* x_t, const double* x_init,
                        const double* m_prev, const double* noise, float sigma_t, double inv_alpha_t, double c_sample, double c_m0,
                        double c_d1, double inv_r0, double c_noise, float inv_resolution, int64_t n_points, int64_t n_per_batch,
                        int32_t scale_batch_column, double* x0_out, float* feats_out, int32_t* coords_out, void* stream);
/* points_to_tensor alone (pipeline:68-84; models.py:162-178 with scale_batch_column = 0): [B, n, 3] points, fp64 (is_f64 != 0)
 * or fp32 -> fp32 features [B n, 3] and int32 voxel coordinates [B n, 4], one launch. */
int lidiff_points_to_field(const void* points, int32_t is_f64, float inv_resolution, int64_t n_points, int64_t n_per_batch,
                           int32_t scale_batch_column, float* feats_out, int32_t* coords_out, void* stream);

/* Farthest-point sampling -- DiffCompletion.preprocess_scan, pipeline:92-105 (open3d farthest_point_down_sample):
 * points [n,3] float64; selected[0] = 0, selected[i+1] = the point farthest (squared distance, first maximum) from
 * selected[0..i].  n_samples - 1 launches on `stream`, no host synchronisation.  workspace: lidiff_fps_workspace_bytes. */
int64_t lidiff_fps_workspace_bytes(int64_t n_points);
int lidiff_fps(const double* points, int64_t n_points, int64_t n_samples, int64_t* selected, void* workspace,
               void* stream);
/* The same selection as ONE persistent cooperative launch (device-wide barrier per selection instead of a launch per
 * selection; 134 ms against 198 ms for 18 000 of 119 035 points).  *status (device int32, zeroed by the call) turns non-zero if the barrier
 * timed out -- the selection is then incomplete: run lidiff_fps.  Returns non-zero, nothing enqueued, when the device
 * cannot hold the grid co-resident (no cooperative launch, or more than 8192 points per CU) -- which
 * lidiff_fps_coop_supported(n_points) answers beforehand (1 / 0), so that a caller never has to interpret an error. */
int32_t lidiff_fps_coop_supported(int64_t n_points);
int lidiff_fps_coop(const double* points, int64_t n_points, int64_t n_samples, int64_t* selected, void* workspace,
                    int32_t* status, void* stream);

/* Nearest neighbour of every point of a [n,3] in b [m,3] (both float when elem_bytes = 4, both double when 8):
 * d2[i] = min_j |a_i - b_j|^2 (element type of the inputs), idx[i] = the lowest such j.  Replaces open3d
 * PointCloud.compute_point_cloud_distance in utils/metrics.py:68,128-129,150-153 (distance = sqrt(d2)) and the
 * K=1 knn_points search inside pytorch3d chamfer_distance, models_refine.py:72.  Exhaustive, exact, deterministic.
 * workspace: lidiff_nn_dist_workspace_bytes(n, m, elem_bytes). */
int64_t lidiff_nn_dist_workspace_bytes(int64_t n, int64_t m, int32_t elem_bytes);
int lidiff_nn_dist(const void* a, int64_t n, const void* b, int64_t m, int32_t elem_bytes, void* d2, int64_t* idx,
                   void* workspace, void* stream);
/* The same search through a uniform grid over b (cubic cells of edge `cell`, in the clouds' unit; binned with the voxel hash of
 * lidiff_vox_unique): a query walks the shells of cells around its own, nearest first, and stops once no unvisited cell can hold
 * a nearer point.  EXACT -- the same d2 bits and the same idx (lowest j on ties) as lidiff_nn_dist, whatever `cell` is; queries far
 * from every point of b (more than 6 shells) and clouds whose cell indices leave +-32767 fall back to the exhaustive scan inside
 * the call.  The refinement network's Chamfer loss (models_refine.py:72: 6 x 180 000 predicted against 2 x 180 000 target points
 * per item) and the evaluation's Chamfer distance (utils/metrics.py:124-141) at full size: ~10^3 distance evaluations per query
 * instead of 3.6 x 10^5.  workspace: lidiff_nn_dist_grid_workspace_bytes(n, m, elem_bytes), 16-byte aligned. */
int64_t lidiff_nn_dist_grid_workspace_bytes(int64_t n, int64_t m, int32_t elem_bytes);
int lidiff_nn_dist_grid(const void* a, int64_t n, const void* b, int64_t m, int32_t elem_bytes, double cell, void* d2, int64_t* idx,
                        void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDIFF_AMD_H */
