"""ctypes binding of oracle/cpp/me_cpu_ref.cpp -- the C++ / OpenMP restatement of MinkowskiEngine's CPU algorithm -- with
the signatures of oracle/me_cpu.py.  TEST INFRASTRUCTURE ONLY (see oracle/me_cpu.py): the fast leg of the checker and the
CPU path timed beside the GPU numbers.  ``oracle.me_cpu.use_cpp()`` routes the oracle's primitives here."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import build as _build

_lib = None
_p, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64


def lib():
    global _lib
    if _lib is None:
        path = _build.LIB if os.path.exists(_build.LIB) else _build.build()
        l = C.CDLL(path)
        l.me_ref_threads.restype = _i32
        l.me_ref_unique.restype = _i32
        l.me_ref_unique.argtypes = [_p, _i64, _i32, _p, _p, _p, _p]
        l.me_ref_kernel_map.argtypes = [_p, _i64, _p, _i64, _i32, _i32, _p]
        l.me_ref_conv_forward.argtypes = [_p, _i64, _i32, _p, _i32, _i32, _p, _i64, _p]
        l.me_ref_argmin_match.argtypes = [_p, _i64, _p, _i64, _p]
        l.me_ref_voxel_mean.argtypes = [_p, _p, _i64, _i32, _i64, _p]
        _lib = l
    return _lib


def threads() -> int:
    return int(lib().me_ref_threads())


def set_threads(n: int) -> None:
    """OpenMP threads of the C++ primitives and of torch-CPU (the oracle's MLP / BatchNorm side)."""
    lib().me_ref_set_threads(int(n))
    torch.set_num_threads(int(n))


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _unique(coords, s):
    c = _np(coords, np.int32).reshape(-1, 4)
    n = c.shape[0]
    uniq = np.empty((n, 4), np.int32)
    inv = np.empty(n, np.int64)
    first = np.empty(n, np.int32)
    m = C.c_int64(0)
    if lib().me_ref_unique(c.ctypes.data, n, s, uniq.ctypes.data, inv.ctypes.data, first.ctypes.data, C.addressof(m)):
        raise ValueError("coordinate outside the 16-bit key range")
    return uniq[:m.value].copy(), inv, first[:m.value].copy()


def voxelize(coords):
    return _unique(coords, 1)


def stride_map(coords, s_out):
    coarse, parent, _ = _unique(coords, int(s_out))
    return coarse, parent.astype(np.int32)


def kernel_map(in_coords, out_coords, ks, ts_in):
    ci, co = _np(in_coords, np.int32), _np(out_coords, np.int32)
    nbr = np.empty((ks ** 3, co.shape[0]), np.int32)
    lib().me_ref_kernel_map(ci.ctypes.data, ci.shape[0], co.ctypes.data, co.shape[0], int(ks), int(ts_in), nbr.ctypes.data)
    return nbr


def conv_forward(feats: torch.Tensor, kernel: torch.Tensor, nbr):
    """fp32 only (the CPU path ME runs); float64 inputs are the numpy oracle's business."""
    x = feats.detach().contiguous().float()
    w = kernel.detach().contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    k, c_in, c_out = w.shape
    m_out = x.shape[0] if nbr is None else nbr.shape[1]
    out = torch.empty(m_out, c_out, dtype=torch.float32)
    nb = None if nbr is None else _np(nbr, np.int32)
    lib().me_ref_conv_forward(x.data_ptr(), x.shape[0], c_in, w.data_ptr(), k, c_out, None if nb is None else nb.ctypes.data,
                              m_out, out.data_ptr())
    return out


def argmin_match(full_c, part_c):
    f, p = _np(full_c, np.int32), _np(part_c, np.int32)
    idx = np.empty(f.shape[0], np.int64)
    lib().me_ref_argmin_match(f.ctypes.data, f.shape[0], p.ctypes.data, p.shape[0], idx.ctypes.data)
    return idx


def voxel_mean(feats: torch.Tensor, inverse, n_vox: int):
    x = feats.detach().contiguous().float()
    inv = _np(inverse, np.int64)
    out = torch.empty(n_vox, x.shape[1], dtype=torch.float32)
    lib().me_ref_voxel_mean(x.data_ptr(), inv.ctypes.data, x.shape[0], x.shape[1], n_vox, out.data_ptr())
    return out
