"""ME.utils.* used by LiDiff."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


def batched_coordinates(coords, dtype=torch.int32, device=None):
    """ME.utils.batched_coordinates (tools/diff_completion_pipeline.py:69, models/models.py:163,
    models_refine.py:33): list of [N_b, D] -> [sum N_b, D+1] with the batch index in column 0.
    Pure tensor plumbing (torch ops), any device."""
    rows = []
    for b, c in enumerate(coords):
        c = torch.as_tensor(c)
        if device is not None:
            c = c.to(device)
        c = c.to(dtype)
        rows.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=dtype, device=c.device), c], dim=1))
    if not rows:
        return torch.zeros((0, 1), dtype=dtype, device=device)
    return torch.cat(rows, dim=0)


def sparse_quantize(coordinates, features=None, return_index=False, return_inverse=False,
                    quantization_size=None, device="cuda"):
    """ME.utils.sparse_quantize (map_from_scans.py:91, SemanticKITTITemporalAggr.py:87,
    utils/pcd_preprocess.py:179): floor to int32, deduplicate voxels; returns the unique
    coordinates (first occurrence order) and, with return_index, the kept row of each.
    Runs the same voxel-hash kernel as TensorField.sparse(); D=3 coordinates (no batch column)."""
    is_np = isinstance(coordinates, np.ndarray)
    c = torch.as_tensor(coordinates)
    if c.dim() != 2 or c.shape[1] != 3:
        raise ValueError("sparse_quantize expects [N, 3] coordinates")
    if c.is_floating_point():
        # divide and floor in the SOURCE dtype (float64 at the map-building call sites, map_from_scans.py:91:
        # metres in the hundreds over 0.1 -- a float32 detour moves points within ~1e-7 relative of a voxel
        # boundary into the neighbouring voxel), on the host when the input lives there
        if quantization_size is not None:
            c = c / quantization_size
        c = torch.floor(c)
    elif quantization_size is not None:
        c = torch.div(c, quantization_size, rounding_mode="floor") if float(quantization_size).is_integer() \
            else torch.floor(c.double() / quantization_size)
    if c.numel() and (c.min() < -32768 or c.max() > 32767):
        raise RuntimeError("coordinate outside the hash-key range [-32768, 32767]")
    ci = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32), c.to(torch.int32).cpu()], dim=1) \
        if not c.is_cuda else torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=c.device),
                                         c.to(torch.int32)], dim=1)
    ci = ci.to(device)
    status = torch.zeros(1, dtype=torch.int32, device=ci.device)
    uniq, inverse, first_idx, _ = ops.vox_unique(ci, status)
    if int(status.item()) != 0:
        raise RuntimeError("coordinate outside the hash-key range")
    out = [uniq[:, 1:]]
    if features is not None:
        out.append(torch.as_tensor(features).to(ci.device)[first_idx.long()])
    if return_index:
        out.append(first_idx.long())
    if return_inverse:
        out.append(inverse)
    if is_np:
        out = [o.cpu().numpy() for o in out]
    return out[0] if len(out) == 1 else tuple(out)
