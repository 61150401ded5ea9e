"""A CPU stand-in for ``import MinkowskiEngine as ME`` built on the oracle (oracle/me_cpu.py).

TEST INFRASTRUCTURE ONLY (see oracle/me_cpu.py): it exists so that the reference's OWN Python --
/root/reference/lidiff/models/minkunet.py and tools/diff_completion_pipeline.py, imported unmodified --
can be executed in the build container, where MinkowskiEngine / pykeops cannot be installed.  Running the
reference's host code over this shim pins everything that IS pinnable without the ME wheel: module tree and
state-dict keys, op order, the ``cat((t, p))`` quirk of minkunet.py:461, batch handling, the pipeline's
tensor plumbing.  The ME-internal arithmetic itself stays restated (PARITY UNPINNED).

``module()`` returns a module object exposing the 12 ME symbols LiDiff consumes;
``keops_module()`` a ``pykeops.torch`` stand-in whose LazyTensor supports exactly the expression of
minkunet.py:412-416.
"""
from __future__ import annotations

import enum
import math
import types

import numpy as np
import torch
import torch.nn as nn

from . import me_cpu as me


class SparseTensorQuantizationMode(enum.Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class MinkowskiAlgorithm(enum.Enum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


class SparseTensor:
    """ME.SparseTensor over an oracle coordinate manager (Appendix A.8 semantics)."""

    def __init__(self, features, tensor_stride, mgr):
        self.F, self.ts, self.mgr = features, int(tensor_stride), mgr

    @property
    def C(self):
        return torch.from_numpy(self.mgr.maps[self.ts])

    @property
    def tensor_stride(self):
        return self.ts

    @property
    def coordinate_manager(self):
        return self.mgr

    def _like(self, f):
        return SparseTensor(f, self.ts, self.mgr)

    def _same(self, other):
        if other.mgr is not self.mgr or other.ts != self.ts:
            raise RuntimeError("sparse tensors live on different coordinate maps")

    def __mul__(self, other):
        if isinstance(other, SparseTensor):
            self._same(other)
            other = other.F
        return self._like(self.F * other)

    def __add__(self, other):
        if isinstance(other, SparseTensor):
            self._same(other)
            other = other.F
        return self._like(self.F + other)

    def slice(self, field):
        if field.mgr is not self.mgr or self.ts != 1:
            raise RuntimeError("slice needs the stride-1 tensor of the field's own coordinate manager")
        out = TensorField.__new__(TensorField)
        out.F = self.F[torch.from_numpy(field.inverse)]
        out.C, out.mgr, out.inverse = field.C, field.mgr, field.inverse
        return out


class TensorField:
    """ME.TensorField (pipeline:74-80); ``sparse()`` = floor, first-occurrence unique, mean (A.3)."""

    def __init__(self, features, coordinates, quantization_mode=SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                 minkowski_algorithm=MinkowskiAlgorithm.DEFAULT, coordinate_manager=None, device=None):
        if quantization_mode is not SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE:
            raise NotImplementedError("LiDiff only uses UNWEIGHTED_AVERAGE")
        self.F = torch.as_tensor(features).cpu()
        self.C = torch.as_tensor(coordinates).cpu()
        self.mgr = coordinate_manager or me.CpuCoordinateManager()
        self.inverse = None

    @property
    def coordinate_manager(self):
        return self.mgr

    def sparse(self):
        ci = me.quantize_floor(self.C.float().numpy())
        uniq, self.inverse, _ = me.voxelize(ci)
        self.mgr.maps[1] = uniq
        return SparseTensor(me.voxel_mean(self.F.float(), self.inverse, uniq.shape[0]), 1, self.mgr)


class _Conv(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=None):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not bias
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = kernel_size, stride
        k_vol = kernel_size ** 3
        shape = (in_channels, out_channels) if k_vol == 1 else (k_vol, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = None
        n = (out_channels if self.transposed else in_channels) * k_vol          # ME's init (Appendix A.6)
        with torch.no_grad():
            self.kernel.uniform_(-1.0 / math.sqrt(n), 1.0 / math.sqrt(n))

    def forward(self, x: SparseTensor):
        cx = me.CpuSparseTensor(x.F, x.ts, x.mgr)
        fn = me.conv_transpose if self.transposed else me.conv
        y = fn(cx, self.kernel, self.kernel_size, self.stride)
        return SparseTensor(y.F, y.ts, y.mgr)


class MinkowskiConvolution(_Conv):
    pass


class MinkowskiConvolutionTranspose(_Conv):
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        return module


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(torch.relu(x.F))


def cat(*tensors):
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    for t in tensors[1:]:
        tensors[0]._same(t)
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))


def _batched_coordinates(coords, dtype=torch.int32, device=None):
    return me.batched_coordinates([torch.as_tensor(c).cpu() for c in coords], dtype=dtype)


def _sparse_quantize(coordinates, features=None, return_index=False, return_inverse=False,
                     quantization_size=None, device=None):
    """ME.utils.sparse_quantize (map_from_scans.py:91; Appendix A.10): floor IN THE INPUT DTYPE, dedup, first
    occurrence order."""
    c = np.asarray(coordinates)
    if quantization_size is not None:
        c = c / quantization_size
    ci = np.floor(c).astype(np.int32)
    ci4 = np.concatenate([np.zeros((ci.shape[0], 1), np.int32), ci], axis=1)
    uniq, inverse, first = me.voxelize(ci4)
    out = [uniq[:, 1:]]
    if features is not None:
        out.append(np.asarray(features)[first])
    if return_index:
        out.append(first.astype(np.int64))
    if return_inverse:
        out.append(inverse)
    return out[0] if len(out) == 1 else tuple(out)


def module() -> types.ModuleType:
    m = types.ModuleType("MinkowskiEngine")
    for obj in (SparseTensorQuantizationMode, MinkowskiAlgorithm, SparseTensor, TensorField, MinkowskiConvolution,
                MinkowskiConvolutionTranspose, MinkowskiBatchNorm, MinkowskiSyncBatchNorm, MinkowskiReLU):
        setattr(m, obj.__name__, obj)
    m.cat = cat
    m.utils = types.ModuleType("MinkowskiEngine.utils")
    m.utils.batched_coordinates = _batched_coordinates
    m.utils.sparse_quantize = _sparse_quantize
    return m


# ----------------------------------------------------------------------------------------
# pykeops.torch.LazyTensor, exactly the expression of minkunet.py:412-416
# ----------------------------------------------------------------------------------------
class LazyTensor:
    """``((LazyTensor(f[:, None, :]) - LazyTensor(p[None, :, :])) ** 2).sum(-1).argKmin(1, dim=1)``: brute-force
    squared distances in the operand dtype, FIRST minimum over j (Appendix A.9)."""

    def __init__(self, x=None, op="leaf", args=()):
        self.x, self.op, self.args = x, op, args

    def __sub__(self, other):
        return LazyTensor(op="sub", args=(self, other))

    def __pow__(self, p):
        assert p == 2
        return LazyTensor(op="sq", args=(self,))

    def sum(self, dim):
        assert dim == -1 and self.op == "sq"
        return LazyTensor(op="sqdist", args=self.args[0].args)

    def argKmin(self, K, dim=1):
        assert K == 1 and dim == 1 and self.op == "sqdist"
        a, b = (t.x for t in self.args)
        assert a.shape[1] == 1 and b.shape[0] == 1
        f, p = a[:, 0, :], b[0]
        idx = torch.empty(f.shape[0], dtype=torch.int64)
        step = max(1, (1 << 24) // max(1, p.shape[0]))
        for s in range(0, f.shape[0], step):
            d = ((f[s:s + step, None, :] - p[None, :, :]) ** 2).sum(-1)
            idx[s:s + step] = torch.argmin(d, dim=1)
        return idx[:, None]


def keops_module():
    top = types.ModuleType("pykeops")
    sub = types.ModuleType("pykeops.torch")
    sub.LazyTensor = LazyTensor
    top.torch = sub
    return top, sub
