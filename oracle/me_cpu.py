"""CPU oracle for the sparse-tensor operator path LiDiff runs through MinkowskiEngine.

TEST INFRASTRUCTURE ONLY.  Nothing in ``lidiff_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do,
and only as the checker / reported CPU baseline.

PARITY UNPINNED: the arithmetic lives in MinkowskiEngine==0.5.4 (pinned only in
/root/reference/README.md:22), pykeops==2.1.2 (requirements.txt:11) and
diffusers==0.18.0 (requirements.txt:12).  None of them is vendored under
/root/reference, installed in this image, or installable (no network), and the reference
ships no tests or golden vectors.  This file therefore restates
the *published* algorithm of those libraries and is anchored on
LiDiff's own call sites, cited per function.  Its independent cross-check is a dense-grid
``torch.nn.functional.conv3d`` on small clouds (tests/test_oracle.py).

Conventions:
  * coordinates are int32 rows ``(b, x, y, z)``;
  * level-0 rows are ordered by first occurrence in point order; strided maps by first
    occurrence when scanning the finer level's rows in order;
  * kernel offsets iterate x fastest: ks=3 -> k = (dx+1) + 3(dy+1) + 9(dz+1) with
    d in {-1,0,1}; ks=2 -> k = dx + 2dy + 4dz with d in {0,1}; the offset (scaled by the
    INPUT tensor stride) is added to the OUTPUT coordinate to find the input voxel;
  * the kernel map is held as a dense neighbour table ``nbr[K, M_out]`` (input row or -1),
    which is equivalent to ME's per-offset (in,out) lists sorted by output row.
"""
from __future__ import annotations

import numpy as np
import torch

KEY_OFF = 32768  # each of (b,x,y,z) must lie in [-32768, 32767]

# Backend of the primitives below: the numpy / torch-CPU restatement in this file, or -- inside ``with use_cpp():`` -- the
# C++ / OpenMP restatement of ME's CPU algorithm (oracle/cpp/me_cpu_ref.cpp through oracle/me_cpp.py: hash-map maps,
# per-offset gather -> MKL SGEMM -> scatter-add).  Both are cross-checked in tests/test_oracle.py; the C++ one is the
# "MinkowskiEngine CPU path" bench.py times beside the GPU numbers (fp32 features only; no autograd).
_CPP = False


class use_cpp:
    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        global _CPP
        self.prev, _CPP = _CPP, self.enabled
        return self

    def __exit__(self, *exc):
        global _CPP
        _CPP = self.prev


# --------------------------------------------------------------------------------------
# coordinates
# --------------------------------------------------------------------------------------
def batched_coordinates(points, dtype=torch.float32):
    """ME.utils.batched_coordinates (call sites: diff_completion_pipeline.py:69,
    models.py:163, models_refine.py:33): list of [N_b, D] -> [sum N_b, D+1], col 0 = b."""
    rows = []
    for b, p in enumerate(points):
        p = torch.as_tensor(p)
        col = torch.full((p.shape[0], 1), b, dtype=dtype)
        rows.append(torch.cat([col, p.to(dtype)], dim=1))
    return torch.cat(rows, dim=0)


def quantize_floor(coords_f: np.ndarray) -> np.ndarray:
    """TensorField.sparse() quantisation: floor of the float field coordinates to int32
    (Appendix A.3; LiDiff rounds first: diff_completion_pipeline.py:72, collations.py:8-12)."""
    return np.floor(np.asarray(coords_f, dtype=np.float32)).astype(np.int32)


def in_key_range(c: np.ndarray) -> np.ndarray:
    c = np.asarray(c, dtype=np.int64)
    # the row whose packed key is all ones is the device table's empty marker: out of range by definition
    return np.all((c >= -KEY_OFF) & (c < KEY_OFF), axis=1) & ~np.all(c == KEY_OFF - 1, axis=1)


def pack_keys(c: np.ndarray) -> np.ndarray:
    """64-bit key, 16 bits per column (offset-binary).  Same packing as the device path."""
    c = np.asarray(c, dtype=np.int64) + KEY_OFF
    return (c[:, 0] << 48) | (c[:, 1] << 32) | (c[:, 2] << 16) | c[:, 3]


def voxelize(coords: np.ndarray):
    """``TensorField.sparse()`` coordinate part (pipeline:149; models.py:99,202;
    minkunet.py:135,597): unique voxel rows in first-occurrence order, the inverse map
    (point -> voxel row, int64) and the index of the first point of every voxel."""
    if _CPP:
        from . import me_cpp
        return me_cpp.voxelize(coords)
    coords = np.asarray(coords, dtype=np.int32)
    if coords.shape[0] == 0:
        return coords.reshape(0, 4), np.zeros(0, np.int64), np.zeros(0, np.int32)
    if not in_key_range(coords).all():
        raise ValueError("coordinate outside the 16-bit key range")
    keys = pack_keys(coords)
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # sorted-unique id -> canonical rank
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    first_idx = first[order].astype(np.int32)
    return coords[first_idx], rank[inv.reshape(-1)].astype(np.int64), first_idx


def voxel_mean(feats: torch.Tensor, inverse: np.ndarray, n_vox: int) -> torch.Tensor:
    """UNWEIGHTED_AVERAGE quantisation (pipeline:77; Appendix A.3): per-voxel mean of the
    member points' features (fp32 accumulation in point order, then divide)."""
    if _CPP:
        from . import me_cpp
        return me_cpp.voxel_mean(feats, inverse, n_vox)
    inv = torch.as_tensor(inverse, dtype=torch.int64)
    out = torch.zeros(n_vox, feats.shape[1], dtype=feats.dtype)
    out.index_add_(0, inv, feats)
    cnt = torch.zeros(n_vox, dtype=feats.dtype)
    cnt.index_add_(0, inv, torch.ones(inv.shape[0], dtype=feats.dtype))
    return out / cnt[:, None]


def floor_to_stride(coords: np.ndarray, s: int) -> np.ndarray:
    out = np.array(coords, dtype=np.int32, copy=True)
    out[:, 1:] = np.floor_divide(out[:, 1:], s) * s   # toward -inf; batch column untouched
    return out


def stride_map(coords: np.ndarray, s_out: int):
    """Coordinate map of a strided convolution (minkunet.py:13-29 used at 103-121,
    184-259, 521-539; Appendix A.5): c_out = floor(c/s_out)*s_out, deduplicated in order
    of first occurrence over the finer rows.  Returns (coarse rows, parent[M_fine])."""
    if _CPP:
        from . import me_cpp
        return me_cpp.stride_map(coords, s_out)
    coarse, parent, _ = voxelize(floor_to_stride(coords, s_out))
    return coarse, parent.astype(np.int32)


def kernel_offsets(ks: int) -> np.ndarray:
    """[K,3] integer offsets, x fastest (Appendix A.6)."""
    if ks == 1:
        return np.zeros((1, 3), np.int32)
    r = np.arange(ks) - (ks - 1) // 2 if ks % 2 == 1 else np.arange(ks)
    dz, dy, dx = np.meshgrid(r, r, r, indexing="ij")
    return np.stack([dx.ravel(), dy.ravel(), dz.ravel()], axis=1).astype(np.int32)


def kernel_map(in_coords: np.ndarray, out_coords: np.ndarray, ks: int, ts_in: int) -> np.ndarray:
    """Kernel map as a neighbour table ``nbr[K, M_out]`` (Appendix A.6):
    nbr[k, o] = row of in_coords equal to out_coords[o] + offset_k * ts_in, else -1.
    ks=3/stride 1: in == out map (minkunet.py:53-66,94,97,156,159,512,515);
    ks=2/stride 2: out = coarse map, in = fine map (minkunet.py:13-29)."""
    if _CPP:
        from . import me_cpp
        return me_cpp.kernel_map(in_coords, out_coords, ks, ts_in)
    in_coords = np.asarray(in_coords, np.int32)
    out_coords = np.asarray(out_coords, np.int32)
    offs = kernel_offsets(ks)
    keys_in = pack_keys(in_coords)
    order = np.argsort(keys_in, kind="stable")
    sk = keys_in[order]
    nbr = np.full((offs.shape[0], out_coords.shape[0]), -1, np.int32)
    for k, off in enumerate(offs):
        q = out_coords.astype(np.int64).copy()
        q[:, 1:] += off.astype(np.int64) * ts_in
        ok = in_key_range(q)
        kq = pack_keys(np.where(ok[:, None], q, 0))
        pos = np.searchsorted(sk, kq)
        pos_c = np.minimum(pos, sk.size - 1) if sk.size else pos
        hit = ok & (pos < sk.size)
        if sk.size:
            hit &= sk[pos_c] == kq
            nbr[k, hit] = order[pos_c[hit]].astype(np.int32)
    return nbr


def transpose_kernel_map(nbr_down: np.ndarray, n_fine: int) -> np.ndarray:
    """Kernel map of MinkowskiConvolutionTranspose(ks=2,s=2) (minkunet.py:32-46): the
    in/out swap of the fine->coarse map, reusing the same kernel index (Appendix A.5)."""
    K, _ = nbr_down.shape
    up = np.full((K, n_fine), -1, np.int32)
    for k in range(K):
        o = np.nonzero(nbr_down[k] >= 0)[0]
        up[k, nbr_down[k, o]] = o.astype(np.int32)
    return up


def rulebook_from_nbr(nbr: np.ndarray):
    """ME-style rulebook: per offset k the (in_row, out_row) pairs sorted by out_row,
    concatenated, plus offset_ptr[K+1] (canonical order of Appendix A.7)."""
    pin, pout, ptr = [], [], [0]
    for k in range(nbr.shape[0]):
        o = np.nonzero(nbr[k] >= 0)[0]
        pin.append(nbr[k, o].astype(np.int32))
        pout.append(o.astype(np.int32))
        ptr.append(ptr[-1] + o.size)
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.int32)
    return cat(pin), cat(pout), np.asarray(ptr, np.int32)


# --------------------------------------------------------------------------------------
# features
# --------------------------------------------------------------------------------------
_BF16_OPERANDS = False


class bf16_operands:
    """``with bf16_operands():`` -- restates the product's mixed-precision TRAINING convolution (BASELINE configs[4],
    include/lidiff_amd.h lidiff_spconv_fwd_bf16) on the CPU: for every convolution whose channel counts are multiples of
    32, forward = conv(bf16(x), bf16(W)), dX = conv^T(bf16(g), bf16(W)), dW = bf16(x)^T bf16(g); sums in the tensors' own
    dtype.  The reference has no such mode of its own (ME has no bf16 kernels): this is the checker of OUR bf16 path."""

    def __enter__(self):
        global _BF16_OPERANDS
        self.prev, _BF16_OPERANDS = _BF16_OPERANDS, True

    def __exit__(self, *exc):
        global _BF16_OPERANDS
        _BF16_OPERANDS = self.prev


def _bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.float().bfloat16().to(t.dtype)


class _Bf16Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, kernel, nbr):
        ctx.save_for_backward(feats, kernel)
        ctx.nbr = nbr
        return _conv_plain(_bf16_round(feats.detach()), _bf16_round(kernel.detach()), nbr)

    @staticmethod
    def backward(ctx, g):
        feats, kernel = ctx.saved_tensors
        with torch.enable_grad():
            xr = _bf16_round(feats.detach()).requires_grad_(True)
            gx, = torch.autograd.grad(_conv_plain(xr, _bf16_round(kernel.detach()), ctx.nbr), xr, _bf16_round(g))
            w = kernel.detach().requires_grad_(True)
            gw, = torch.autograd.grad(_conv_plain(_bf16_round(feats.detach()), w, ctx.nbr), w, _bf16_round(g))
        return gx, gw, None


def conv_forward(feats: torch.Tensor, kernel: torch.Tensor, nbr: np.ndarray | None) -> torch.Tensor:
    """MinkowskiConvolution / ConvolutionTranspose forward, ME CPU algorithm (Appendix
    A.6): for k ascending: gather rows -> buf @ W[k] -> out[out_row] += res.
    ``kernel`` is [K,Cin,Cout], or [Cin,Cout] for kernel_size=1 (``F.mm(kernel)``,
    minkunet.py:72)."""
    if _BF16_OPERANDS and kernel.shape[-2] % 32 == 0 and kernel.shape[-1] % 32 == 0:
        return _Bf16Conv.apply(feats, kernel, nbr)
    return _conv_plain(feats, kernel, nbr)


def _conv_plain(feats: torch.Tensor, kernel: torch.Tensor, nbr: np.ndarray | None) -> torch.Tensor:
    if _CPP and feats.dtype == torch.float32 and not (feats.requires_grad or kernel.requires_grad):
        from . import me_cpp
        return me_cpp.conv_forward(feats, kernel, nbr)
    if kernel.dim() == 2:
        return feats @ kernel
    K, _, cout = kernel.shape
    out = torch.zeros(nbr.shape[1], cout, dtype=feats.dtype)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        if o.size == 0:
            continue
        i = torch.from_numpy(nbr[k, o].astype(np.int64))
        out.index_add_(0, torch.from_numpy(o.astype(np.int64)), feats[i] @ kernel[k])
    return out


def batch_norm_eval(x, weight, bias, mean, var, eps=1e-5):
    """MinkowskiBatchNorm in eval mode = nn.BatchNorm1d on F (Appendix A.8)."""
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def argmin_match(full_c: np.ndarray, part_c: np.ndarray) -> np.ndarray:
    """MinkUNetDiff.match_part_to_full (minkunet.py:403-418): for every full voxel the
    index of the nearest part voxel by squared L2 over (b*2*max_coord, x, y, z); ties go
    to the lowest index (KeOps argKmin, Appendix A.9).  Exact integer arithmetic (the
    reference's fp32 is exact for |c| < 2^12, which LiDiff's coordinates satisfy)."""
    if _CPP:
        from . import me_cpp
        return me_cpp.argmin_match(full_c, part_c)
    f = np.asarray(full_c, np.int64).copy()
    p = np.asarray(part_c, np.int64).copy()
    scale = int(f.max()) * 2
    f[:, 0] *= scale
    p[:, 0] *= scale
    # brute force as one GEMM per chunk: |f|^2 + |p|^2 - 2 f.p is exact in floating point while
    # every intermediate integer stays below the mantissa range (checked, float64 otherwise); torch.argmin returns the FIRST minimum, i.e. the lowest index on ties.
    bound = 4 * (int(np.abs(f).max()) + int(np.abs(p).max())) ** 2     # every intermediate below this
    dt = torch.float32 if bound < (1 << 24) else torch.float64           # float32 is exact below 2^24
    ft, pt = torch.from_numpy(f).to(dt), torch.from_numpy(p).to(dt)
    p2 = (pt * pt).sum(1)[None, :]
    idx = torch.empty(ft.shape[0], dtype=torch.int64)
    step = max(1, (1 << 25) // max(1, pt.shape[0]))
    for s in range(0, ft.shape[0], step):
        fc = ft[s:s + step]
        d = (fc * fc).sum(1)[:, None] + p2 - 2.0 * (fc @ pt.t())
        idx[s:s + step] = torch.argmin(d, dim=1)
    return idx.numpy()


# --------------------------------------------------------------------------------------
# a minimal coordinate manager + sparse tensor, enough for the networks in minkunet_cpu.py
# --------------------------------------------------------------------------------------
class CpuCoordinateManager:
    """Caches coordinate maps by tensor stride and kernel maps by (ts_in, ts_out, ks,
    transposed) exactly as LiDiff relies on (SURVEY.md 8b 'Ownership / lifetime')."""

    def __init__(self):
        self.maps: dict[int, np.ndarray] = {}
        self.parents: dict[int, np.ndarray] = {}
        self.kmaps: dict[tuple, np.ndarray] = {}

    def stride(self, ts: int, s: int) -> int:
        ts_out = ts * s
        if ts_out not in self.maps:
            self.maps[ts_out], self.parents[ts_out] = stride_map(self.maps[ts], ts_out)
        return ts_out

    def kernel_map(self, ts_in: int, ts_out: int, ks: int, transposed=False) -> np.ndarray:
        key = (ts_in, ts_out, ks, transposed)
        if key not in self.kmaps:
            if transposed:      # in = coarse (ts_in), out = fine (ts_out)
                down = self.kernel_map(ts_out, ts_in, ks)
                self.kmaps[key] = transpose_kernel_map(down, self.maps[ts_out].shape[0])
            else:
                self.kmaps[key] = kernel_map(self.maps[ts_in], self.maps[ts_out], ks, ts_in)
        return self.kmaps[key]


class CpuSparseTensor:
    def __init__(self, F: torch.Tensor, ts: int, mgr: CpuCoordinateManager):
        self.F, self.ts, self.mgr = F, ts, mgr

    @property
    def C(self):
        return self.mgr.maps[self.ts]

    def replace(self, F):
        return CpuSparseTensor(F, self.ts, self.mgr)


class CpuTensorField:
    """ME.TensorField + .sparse() + slice target (pipeline:74-80; Appendix A.2/A.3)."""

    def __init__(self, features: torch.Tensor, coordinates: torch.Tensor):
        self.F = features
        self.coords_f = coordinates
        self.mgr = CpuCoordinateManager()
        self.inverse = None

    def sparse(self) -> CpuSparseTensor:
        ci = quantize_floor(self.coords_f.numpy())
        uniq, self.inverse, _ = voxelize(ci)
        self.mgr.maps[1] = uniq
        return CpuSparseTensor(voxel_mean(self.F, self.inverse, uniq.shape[0]), 1, self.mgr)


def conv(x: CpuSparseTensor, kernel: torch.Tensor, ks: int, stride: int) -> CpuSparseTensor:
    if ks == 1:
        return x.replace(conv_forward(x.F, kernel, None))
    ts_out = x.mgr.stride(x.ts, stride) if stride > 1 else x.ts
    nbr = x.mgr.kernel_map(x.ts, ts_out, ks)
    return CpuSparseTensor(conv_forward(x.F, kernel, nbr), ts_out, x.mgr)


def conv_transpose(x: CpuSparseTensor, kernel: torch.Tensor, ks: int, stride: int) -> CpuSparseTensor:
    ts_out = x.ts // stride
    assert ts_out in x.mgr.maps, "transposed conv needs the encoder's finer map (A.5)"
    nbr = x.mgr.kernel_map(x.ts, ts_out, ks, transposed=True)
    return CpuSparseTensor(conv_forward(x.F, kernel, nbr), ts_out, x.mgr)
