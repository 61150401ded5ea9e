"""``pykeops.torch.LazyTensor`` for the one expression LiDiff writes with it
(/root/reference/lidiff/models/minkunet.py:412-416):

    f = LazyTensor(full_c[:, None, :]); p = LazyTensor(part_c[None, :, :])
    idx = ((f - p) ** 2).sum(-1).argKmin(1, dim=1)[:, 0]

The symbolic tree is kept lazily (as KeOps does) and ``argKmin(1, dim=1)`` runs the HIP brute-force arg-min
``lidiff_argmin_rows_f32`` (fp32, lowest index on ties: SURVEY.md Appendix A.9).  Anything else raises: this is
the pykeops surface of LiDiff, not a KeOps implementation.  GPU only (no CPU fallback).
"""
from __future__ import annotations

import torch

from .. import ops


class LazyTensor:
    def __init__(self, x: torch.Tensor | None = None, _op: str = "var", _args=()):
        if _op == "var":
            if not isinstance(x, torch.Tensor) or x.dim() != 3 or 1 not in (x.shape[0], x.shape[1]):
                raise NotImplementedError("LazyTensor shim: variables must be [M,1,D] or [1,N,D] tensors")
        self.x, self.op, self.args = x, _op, _args

    def __sub__(self, other):
        if not isinstance(other, LazyTensor) or self.op != "var" or other.op != "var":
            raise NotImplementedError("LazyTensor shim: only (variable - variable)")
        return LazyTensor(_op="sub", _args=(self, other))

    def __pow__(self, p):
        if p != 2 or self.op != "sub":
            raise NotImplementedError("LazyTensor shim: only (x - y) ** 2")
        return LazyTensor(_op="sq", _args=self.args)

    def sum(self, dim=-1):
        if dim not in (-1, 2) or self.op != "sq":
            raise NotImplementedError("LazyTensor shim: only ((x - y) ** 2).sum(-1)")
        return LazyTensor(_op="sqdist", _args=self.args)

    def argKmin(self, K, dim=1):
        if K != 1 or self.op != "sqdist":
            raise NotImplementedError("LazyTensor shim: only sqdist.argKmin(1, dim=...)")
        a, b = (t.x for t in self.args)
        if a.shape[1] == 1 and b.shape[0] == 1:            # a indexed by i, b by j
            rows_i, rows_j = a[:, 0, :], b[0]
        elif a.shape[0] == 1 and b.shape[1] == 1:
            rows_i, rows_j = b[:, 0, :], a[0]
        else:
            raise NotImplementedError("LazyTensor shim: one [M,1,D] and one [1,N,D] variable")
        if dim == 0:                                        # reduce over i: nearest i for every j
            rows_i, rows_j = rows_j, rows_i
        elif dim != 1:
            raise NotImplementedError("dim must be 0 or 1")
        return ops.argmin_rows(rows_i, rows_j)[:, None]
