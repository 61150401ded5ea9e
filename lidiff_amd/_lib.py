"""ctypes binding of liblidiff_amd.so (the C ABI declared in include/lidiff_amd.h).

There is no CPU fallback: if the shared library is missing or a call is rejected, this
module raises.  Torch is only plumbing here -- it owns device memory and the HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblidiff_amd.so")

_p, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64

# name -> (restype, argtypes); mirrors include/lidiff_amd.h one to one
SIGNATURES = {
    "lidiff_abi_version": (_i32, []),
    "lidiff_last_error": (C.c_char_p, []),
    "lidiff_hash_capacity": (_i64, [_i64]),
    "lidiff_unique_workspace_bytes": (_i64, [_i64]),
    "lidiff_coords_floor": (_i32, [_p, _i64, _p, _p]),
    "lidiff_vox_unique": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _i32, _p]),
    "lidiff_vox_mean_workspace_bytes": (_i64, [_i64, _i32]),
    "lidiff_vox_mean": (_i32, [_p, _p, _i64, _i32, _i64, _p, _p, _p, _i32, _p]),
    "lidiff_vox_mean_bwd": (_i32, [_p, _p, _p, _i64, _i32, _p, _p]),
    "lidiff_map_stride": (_i32, [_p, _i64, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "lidiff_kernel_map": (_i32, [_p, _i64, _p, _p, _i64, _i32, _i32, _p, _p]),
    "lidiff_kernel_map_self": (_i32, [_p, _i64, _p, _p, _i64, _i32, _p, _p]),
    "lidiff_kernel_map_self_dev": (_i32, [_p, _i64, _p, _p, _p, _i64, _i32, _p, _i32, _p]),
    "lidiff_map_stride_dev": (_i32, [_p, _i64, _p, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _i32, _p]),
    "lidiff_tail_map_dev": (_i32, [_p, _i32, _i64, _p, _i32, _p, _p, _i64, _p, _p, _p, _p]),
    "lidiff_kernel_map_down": (_i32, [_p, _p, _i64, _i32, _i64, _p, _p]),
    "lidiff_kernel_map_up": (_i32, [_p, _p, _i64, _i32, _p, _p]),
    "lidiff_rulebook_compact": (_i32, [_p, _i32, _i64, _p, _p, _p, _p, _p]),
    "lidiff_rulebook_workspace_bytes": (_i64, [_i32, _i64]),
    "lidiff_tail_map_workspace_bytes": (_i64, [_i32, _i64]),
    "lidiff_tail_map": (_i32, [_p, _i32, _i64, _i32, _p, _p, _i64, _p, _p, _p, _p]),
    "lidiff_spconv_packed_weight_floats": (_i64, [_i32, _i32, _i32]),
    "lidiff_spconv_pack_weights": (_i32, [_p, _i32, _i32, _i32, _p, _p]),
    "lidiff_spconv_fwd": (_i32, [_p, _i32, _p, _i32, _p, _p, _i32, _i64, _i64, _i32, _p, _p, _p, _p, _i32, _p, _i32, _i32,
                                 _p, _p, _p, _i64, _p, _p]),
    "lidiff_spconv_fwd_kernel_id": (_i32, [_i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "lidiff_spconv_fwd_pairs_supported": (_i32, [_i32, _i32, _i32]),
    "lidiff_spconv_fwd_pairs": (_i32, [_p, _i32, _p, _i32, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _i32,
                                       _i32, _p]),
    "lidiff_spconv_packed_weight_bf16_elems": (_i64, [_i32, _i32, _i32, _i32]),
    "lidiff_spconv_pack_weights_bf16": (_i32, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "lidiff_spconv_fwd_bf16": (_i32, [_p, _i32, _p, _i32, _p, _i32, _p, _i32, _i64, _i64, _i32, _p, _p, _p, _p, _i32, _i32, _i32, _p]),
    "lidiff_cast_bf16": (_i32, [_p, _i64, _p, _p]),
    "lidiff_split3_rows": (_i32, [_p, _i64, _i32, _i32, _p, _p, _p, _i64, _p]),
    "lidiff_spconv_fwd_split3_supported": (_i32, [_i32, _i32, _i32]),
    "lidiff_spconv_fwd_split3": (_i32, [_p, _i32, _p, _i32, _p, _p, _i32, _i64, _i64, _i32, _p, _p, _p, _p, _p, _i32, _i32, _p, _p, _i32, C.c_float, _p, _p]),
    "lidiff_spconv_pack_weights_f16x2": (_i32, [_p, _i32, _i32, _i32, C.c_float, _p, _p, _p]),
    "lidiff_row_mask_keys": (_i32, [_p, _i32, _i64, _p, _p]),
    "lidiff_spconv_bwd_w_workspace_floats": (_i64, [_i32, _i32, _i32, _i64]),
    "lidiff_spconv_bwd_w": (_i32, [_p, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i32, _p, _p, _p]),
    "lidiff_spconv_bwd_w_bf16": (_i32, [_p, _i32, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i32, _p, _p, _i32, _p]),
    "lidiff_bn_workspace_bytes": (_i64, [_i32]),
    "lidiff_bn_stats": (_i32, [_p, _i64, _i32, C.c_float, _p, _p, _p, _p, _p, C.c_float, _p, _p]),
    "lidiff_bn_apply": (_i32, [_p, _i64, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p]),
    "lidiff_bn_bwd": (_i32, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lidiff_cfg_dpm_step": (_i32, [_p, _p, C.c_float, _p, _p, _p, _p, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_float, _i64, _i64, _i32, _p, _p, _p, _p]),
    "lidiff_points_to_field": (_i32, [_p, _i32, C.c_float, _i64, _i64, _i32, _p, _p, _p]),
    "lidiff_bn_sums": (_i32, [_p, _i64, _i32, _p, _p, _p]),
    "lidiff_bn_stats_from_sums": (_i32, [_p, _i32, C.c_float, _p, _p, _p, _p, _p, C.c_float, _p]),
    "lidiff_bn_bwd_sums": (_i32, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p]),
    "lidiff_bn_bwd_apply": (_i32, [_p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lidiff_morton_keys": (_i32, [_p, _i64, _i32, _p, _p]),
    "lidiff_gather_rows": (_i32, [_p, _p, _i64, _i32, _p, _p]),
    "lidiff_slice_head_supported": (_i32, [_i32, _i32, _i32]),
    "lidiff_slice_head": (_i32, [_p, _p, _i64, _i64, _i32, _i32, _p, _p, _i32, _p, _p, _i32, C.c_float, _p, _p]),
    "lidiff_gather_bias_leaky": (_i32, [_p, _p, _p, _i64, _i32, C.c_float, _p, _p]),
    "lidiff_segment_sum_workspace_bytes": (_i64, [_i64, _i32]),
    "lidiff_segment_sum_rows": (_i32, [_p, _p, _p, _i64, _i32, _p, _i64, _p, _p]),
    "lidiff_gather_mul_rows": (_i32, [_p, _p, _p, _i64, _i32, _p, _p, _p]),
    "lidiff_nn_match": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _p]),
    "lidiff_nn_match_dev": (_i32, [_p, _i64, _p, _p, _i64, _p, _p, _p]),
    "lidiff_argmin_rows_f32": (_i32, [_p, _i64, _p, _i64, _p, _p]),
    "lidiff_fps_workspace_bytes": (_i64, [_i64]),
    "lidiff_fps": (_i32, [_p, _i64, _i64, _p, _p, _p]),
    "lidiff_fps_coop_supported": (_i32, [_i64]),
    "lidiff_fps_coop": (_i32, [_p, _i64, _i64, _p, _p, _p, _p]),
    "lidiff_kernel_map_down_dev": (_i32, [_p, _p, _i64, _p, _i32, _i64, _p, _i32, _p]),
    "lidiff_kernel_map_up_dev": (_i32, [_p, _p, _i64, _p, _i32, _p, _p]),
    "lidiff_tail_map_fill_bounded": (_i32, [_p, _i32, _i64, _p, _i32, _p, _p, _i64, _p, _p, _p, _p, _p]),
    "lidiff_publish_words": (_i32, [_p, _i32, _p, _p, _i32, _p]),
    "lidiff_host_device_pointer": (_i32, [_p, _p]),
    "lidiff_nn_dist_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "lidiff_nn_dist": (_i32, [_p, _i64, _p, _i64, _i32, _p, _p, _p, _p]),
    "lidiff_nn_dist_grid_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "lidiff_nn_dist_grid": (_i32, [_p, _i64, _p, _i64, _i32, C.c_double, _p, _p, _p, _p]),
}

ABI_VERSION = 28
_lib = None


def load() -> C.CDLL:
    """Load the HIP extension; raises if it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"lidiff_amd HIP extension missing: {LIB_PATH}. Build it with "
                "`python -m lidiff_amd.csrc.build` (needs hipcc); there is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.lidiff_abi_version() != ABI_VERSION:
            raise RuntimeError("lidiff_amd ABI version mismatch")
        _lib = lib
    return _lib


# --- asynchronous map building (CoordinateManager.set_async): coordinate maps are built on side streams while the caller's
# stream runs the convolutions that are already queued.  Every launch through call() on a stream that is NOT a build stream first
# joins the build streams that have queued work since the last join (one event each) -- so a consumer can never run ahead of the
# map it reads, whichever operator it goes through.
# CONTRACT: a tensor built under CoordinateManager.building() may be consumed (a) through call() -- every entry point joins first --
# or (b) by a torch op only behind a host read of the same build stream (the map sizes: build_pyramid's one read) or behind
# DiffCompletion._adopt / an explicit event; nothing else touches them (TailMap.fill, up_order and the voxel mean all run inside
# building() themselves and hand their results on through call()).
_BUILD_STREAMS: set = set()
_PENDING: dict = {}          # build stream -> None (join everything queued on it so far) | an event (join up to that event only)


def register_build_stream(stream) -> None:
    _BUILD_STREAMS.add(stream)


def mark_pending(stream, upto=None) -> None:
    """upto: an event already recorded on `stream` -- consumers need only what was queued before it (ops.build_pyramid_lanes: the
    stem waits for level 0, not for the whole pyramid).  A later mark without an event widens the join to the whole stream again."""
    _PENDING[stream] = upto


def unmark_pending(stream) -> None:
    """The work queued on `stream` is handed over through events of its own (CoordinateManager._acquire): no blanket join."""
    _PENDING.pop(stream, None)


def join_pending() -> None:
    if not _PENDING:
        return
    cur = torch.cuda.current_stream()
    if cur in _BUILD_STREAMS:
        return
    for st, ev in list(_PENDING.items()):
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(st)
        cur.wait_event(ev)
    _PENDING.clear()


def ptr(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args):
    """Invoke an int-returning entry point; non-zero status -> RuntimeError (ME raises too)."""
    if _PENDING:
        join_pending()
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {load().lidiff_last_error().decode()}")


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lidiff_amd operators run on the GPU only (no CPU fallback); "
                               f"got a {t.device} tensor")
