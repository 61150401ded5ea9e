"""Imports the reference's OWN Python -- /root/reference/lidiff/models/minkunet.py and
tools/diff_completion_pipeline.py, byte for byte as they lie there -- with its un-installable third-party imports
(MinkowskiEngine, pykeops, diffusers, open3d, pytorch_lightning, natsort) resolved to stand-ins:

  * ``backend="oracle"``: the CPU oracle shim (oracle/me_shim.py) -- runs in the build container, no GPU;
  * ``backend="hip"``   : the product (lidiff_amd.compat.install()) -- the HIP kernels behind ME's API.

Nothing of the reference is copied or edited; the files are read from /root/reference at test time (absent on the
GPU box: those tests skip there, the fixtures they generate travel instead -- tests/golden/make_golden.py).
Two module GLOBALS of the imported copy are rebound after import (not the file): ``torch`` becomes a proxy whose
``device('cuda')`` answers with the test's device, because minkunet.py:395 hard-codes ``torch.device('cuda')``.
"""
from __future__ import annotations

import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = "/root/reference"
REF_PKG = os.path.join(REF_ROOT, "lidiff")


def have_reference() -> bool:
    return os.path.exists(os.path.join(REF_PKG, "models", "minkunet.py"))


class _TorchProxy:
    """``torch`` with ``device('cuda')`` redirected (minkunet.py:395)."""

    def __init__(self, device):
        self._device = torch.device(device)

    def __getattr__(self, name):
        return getattr(torch, name)

    def device(self, *args, **kwargs):
        d = torch.device(*args, **kwargs)
        return self._device if d.type == "cuda" else d


@contextlib.contextmanager
def _aliases(mods: dict):
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _backend_modules(backend: str):
    if backend == "oracle":
        from oracle import me_shim
        me = me_shim.module()
        ktop, ksub = me_shim.keops_module()
        from lidiff_amd.compat import diffusers_alias as diff      # host-only scheduler, device agnostic
    else:
        import lidiff_amd.MinkowskiEngine as me
        from lidiff_amd.compat import diffusers_alias as diff
        from lidiff_amd.compat import keops as ksub
        ktop = types.ModuleType("pykeops")
        ktop.torch = ksub
    return {"MinkowskiEngine": me, "MinkowskiEngine.utils": me.utils, "pykeops": ktop, "pykeops.torch": ksub,
            "diffusers": diff}


def _load(path, name, mods, device):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    with _aliases(mods):
        sys.modules[name] = mod
        try:
            spec.loader.exec_module(mod)
        finally:
            sys.modules.pop(name, None)
    mod.torch = _TorchProxy(device)
    return mod


def reference_minkunet(backend: str, device="cpu"):
    """The reference's models/minkunet.py as a module object (classes MinkGlobalEnc, MinkUNetDiff, MinkUNet)."""
    mods = _backend_modules(backend)
    return _load(os.path.join(REF_PKG, "models", "minkunet.py"), f"_ref_minkunet_{backend}", mods, device), mods


# ----------------------------------------------------------------------------------------
# stand-ins for the pipeline file's other imports (tools/diff_completion_pipeline.py:1-13)
# ----------------------------------------------------------------------------------------
class _LightningModule(nn.Module):
    """pytorch_lightning.core.lightning.LightningModule: what DiffCompletion uses of it (pipeline:15-56):
    save_hyperparameters -> self.hparams (a dict), .device, nn.Module behaviour."""

    def __init__(self):
        super().__init__()
        self._hparams = {}

    def save_hyperparameters(self, hp):
        self._hparams = {k: (dict(v) if isinstance(v, dict) else v) for k, v in dict(hp).items()}

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")


def _fps_float64(pts: np.ndarray, n: int) -> np.ndarray:
    """open3d PointCloud.farthest_point_down_sample (open3d==0.17.0, un-installable here): greedy from index 0,
    float64 squared distances, first maximum -- oracle/fps_cpu.py restates it."""
    from oracle.fps_cpu import farthest_point_sample
    return farthest_point_sample(pts, n)


def _open3d_stub():
    o3d = types.ModuleType("open3d")

    class Vector3dVector(np.ndarray):
        def __new__(cls, a):
            return np.asarray(a, dtype=np.float64).view(cls)

    class PointCloud:
        def __init__(self):
            self.points = np.zeros((0, 3))

        def farthest_point_down_sample(self, n):
            out = PointCloud()
            pts = np.asarray(self.points)
            out.points = pts[_fps_float64(pts, n)]
            return out

        def estimate_normals(self):
            pass

        def compute_point_cloud_distance(self, other):
            """open3d: per point of self the Euclidean distance to its nearest point of `other` (brute force, float64)."""
            a, b = np.asarray(self.points, np.float64), np.asarray(other.points, np.float64)
            out = np.empty(a.shape[0])
            for lo in range(0, a.shape[0], 512):
                out[lo:lo + 512] = np.sqrt(((a[lo:lo + 512, None, :] - b[None, :, :]) ** 2).sum(-1).min(axis=1))
            return out

    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=Vector3dVector)
    o3d.io = types.SimpleNamespace(read_point_cloud=None, write_point_cloud=None)
    return o3d


def reference_pipeline(backend: str, device="cpu"):
    """The reference's tools/diff_completion_pipeline.py as a module object (class DiffCompletion), importing the
    reference's own lidiff.models.minkunet."""
    minknet, mods = reference_minkunet(backend, device)
    pl_top = types.ModuleType("pytorch_lightning")
    pl_core = types.ModuleType("pytorch_lightning.core")
    pl_light = types.ModuleType("pytorch_lightning.core.lightning")
    pl_light.LightningModule = _LightningModule
    pl_top.core, pl_core.lightning = pl_core, pl_light
    pl_top.LightningModule = _LightningModule
    natsort = types.ModuleType("natsort")
    natsort.natsorted = sorted
    lidiff_pkg = types.ModuleType("lidiff")
    lidiff_models = types.ModuleType("lidiff.models")
    lidiff_pkg.models, lidiff_models.minkunet = lidiff_models, minknet
    mods = dict(mods)
    mods.update({"open3d": _open3d_stub(), "pytorch_lightning": pl_top, "pytorch_lightning.core": pl_core,
                 "pytorch_lightning.core.lightning": pl_light, "natsort": natsort, "lidiff": lidiff_pkg,
                 "lidiff.models": lidiff_models, "lidiff.models.minkunet": minknet})
    return _load(os.path.join(REF_PKG, "tools", "diff_completion_pipeline.py"), f"_ref_pipeline_{backend}", mods,
                 device), minknet


def reference_models(backend: str, device="cpu"):
    """The reference's models/models.py as a module object (class DiffusionPoints: the Lightning training module,
    models.py:18-217), importing the reference's own lidiff.models.minkunet and lidiff.utils.{scheduling, collations,
    metrics} -- all read from /root/reference -- over the stand-ins."""
    minknet, mods = reference_minkunet(backend, device)

    class _Lightning(_LightningModule):
        def log(self, *a, **k):
            pass

    pl_top = types.ModuleType("pytorch_lightning")
    pl_core = types.ModuleType("pytorch_lightning.core")
    pl_light = types.ModuleType("pytorch_lightning.core.lightning")
    pl_light.LightningModule = _Lightning
    pl_top.core, pl_core.lightning = pl_core, pl_light
    pl_top.LightningModule = _Lightning
    pl_top.LightningDataModule = type("LightningDataModule", (), {})
    o3d = _open3d_stub()
    o3d.geometry.Geometry = type("Geometry", (), {})
    mods = dict(mods)
    mods.update({"open3d": o3d, "pytorch_lightning": pl_top, "pytorch_lightning.core": pl_core,
                 "pytorch_lightning.core.lightning": pl_light})
    utils = {}
    for name in ("scheduling", "collations", "metrics"):
        utils[name] = _load(os.path.join(REF_PKG, "utils", name + ".py"), f"_ref_utils_{name}_{backend}", mods, device)
    lidiff_pkg, lidiff_models, lidiff_utils = (types.ModuleType("lidiff"), types.ModuleType("lidiff.models"),
                                               types.ModuleType("lidiff.utils"))
    lidiff_pkg.models, lidiff_pkg.utils, lidiff_models.minkunet = lidiff_models, lidiff_utils, minknet
    mods.update({"lidiff": lidiff_pkg, "lidiff.models": lidiff_models, "lidiff.models.minkunet": minknet,
                 "lidiff.utils": lidiff_utils})
    for name, mod in utils.items():
        setattr(lidiff_utils, name, mod)
        mods["lidiff.utils." + name] = mod
    return _load(os.path.join(REF_PKG, "models", "models.py"), f"_ref_models_{backend}", mods, device), minknet


def _chamfer_distance_stub(x, y):
    """pytorch3d.loss.chamfer_distance with its defaults (pytorch3d is not installable here): for [B,N,3] vs [B,M,3] the
    squared distance of every point to its nearest neighbour in the other cloud, mean over points, both directions
    added, batch mean.  Brute force in torch (differentiable through the distances, as pytorch3d's knn_gather is)."""
    d = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)                 # [B, N, M]
    return (d.min(dim=2).values.mean(dim=1) + d.min(dim=1).values.mean(dim=1)).mean(), None


def reference_models_refine(backend: str, device="cpu"):
    """The reference's models/models_refine.py (class RefineDiffusion, models_refine.py:18-76) over the stand-ins."""
    minknet, mods = reference_minkunet(backend, device)
    pl_light = types.ModuleType("pytorch_lightning.core.lightning")

    class _Lightning(_LightningModule):
        def log(self, *a, **k):
            pass

    pl_top, pl_core = types.ModuleType("pytorch_lightning"), types.ModuleType("pytorch_lightning.core")
    pl_light.LightningModule = _Lightning
    pl_top.core, pl_core.lightning = pl_core, pl_light
    pl_top.LightningModule = _Lightning
    pl_top.LightningDataModule = type("LightningDataModule", (), {})
    o3d = _open3d_stub()
    o3d.geometry.Geometry = type("Geometry", (), {})
    p3d, p3d_loss = types.ModuleType("pytorch3d"), types.ModuleType("pytorch3d.loss")
    p3d_loss.chamfer_distance = _chamfer_distance_stub
    p3d.loss = p3d_loss
    mods = dict(mods)
    mods.update({"open3d": o3d, "pytorch_lightning": pl_top, "pytorch_lightning.core": pl_core,
                 "pytorch_lightning.core.lightning": pl_light, "pytorch3d": p3d, "pytorch3d.loss": p3d_loss})
    utils = {name: _load(os.path.join(REF_PKG, "utils", name + ".py"), f"_ref_utils2_{name}_{backend}", mods, device)
             for name in ("scheduling", "collations", "metrics")}
    lidiff_pkg, lidiff_models, lidiff_utils = (types.ModuleType("lidiff"), types.ModuleType("lidiff.models"),
                                               types.ModuleType("lidiff.utils"))
    lidiff_pkg.models, lidiff_pkg.utils, lidiff_models.minkunet = lidiff_models, lidiff_utils, minknet
    mods.update({"lidiff": lidiff_pkg, "lidiff.models": lidiff_models, "lidiff.models.minkunet": minknet,
                 "lidiff.utils": lidiff_utils})
    for name, mod in utils.items():
        setattr(lidiff_utils, name, mod)
        mods["lidiff.utils." + name] = mod
    return _load(os.path.join(REF_PKG, "models", "models_refine.py"), f"_ref_models_refine_{backend}", mods, device), minknet


def reference_metrics():
    """The reference's utils/metrics.py (RMSE, ChamferDistance, CompletionIoU, PrecisionRecall) over the open3d stand-in."""
    o3d = _open3d_stub()
    o3d.geometry.Geometry = type("Geometry", (), {})
    return _load(os.path.join(REF_PKG, "utils", "metrics.py"), "_ref_metrics", {"open3d": o3d}, "cpu"), o3d


@contextlib.contextmanager
def cuda_calls_as(device):
    """The reference writes ``.cuda()`` (pipeline:23-24,34,59-66,100); inside this context those calls move to
    `device` instead (a patch of torch for the duration of a test, not of the reference)."""
    dev = torch.device(device)
    saved = (torch.Tensor.cuda, nn.Module.cuda, torch.cuda.empty_cache)
    torch.Tensor.cuda = lambda self, *a, **k: self.to(dev)
    nn.Module.cuda = lambda self, *a, **k: self.to(dev)
    torch.cuda.empty_cache = lambda: None
    try:
        yield
    finally:
        torch.Tensor.cuda, nn.Module.cuda, torch.cuda.empty_cache = saved
