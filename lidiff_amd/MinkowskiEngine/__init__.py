"""Drop-in for the subset of the MinkowskiEngine 0.5.4 Python API that LiDiff consumes
(SURVEY.md 8b): ``import lidiff_amd.MinkowskiEngine as ME``.

Symbols (reference call sites, paths relative to /root/reference/lidiff):
  ME.utils.batched_coordinates            tools/diff_completion_pipeline.py:69, models/models.py:163
  ME.utils.sparse_quantize                map_from_scans.py:91, SemanticKITTITemporalAggr.py:87
  ME.TensorField(.F .C .sparse())         pipeline:74-80,149; models.py:168-174
  ME.SparseTensor(.F .C * + .slice)       models/minkunet.py:431,79,497
  ME.MinkowskiConvolution[Transpose]      minkunet.py:17,36,53,61,72
  ME.MinkowskiBatchNorm / SyncBatchNorm   minkunet.py:23; train.py:90
  ME.MinkowskiReLU, ME.cat                minkunet.py:24,464
  ME.SparseTensorQuantizationMode, ME.MinkowskiAlgorithm   pipeline:77-78

Everything runs on the GPU through the C ABI in include/lidiff_amd.h (hand-written HIP
kernels); there is no CPU path.  Row order is deterministic: first occurrence in point
order (level 0) / in finer-row order (strided maps) -- SURVEY.md Appendix A.7.
"""
from __future__ import annotations

import contextlib
import enum
import math

import torch
import torch.nn as nn

from .. import ops
from . import utils  # noqa: F401  (ME.utils.*)


class SparseTensorQuantizationMode(enum.Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1
    UNWEIGHTED_SUM = 2
    NO_QUANTIZATION = 3


class MinkowskiAlgorithm(enum.Enum):
    DEFAULT = 0
    MEMORY_EFFICIENT = 1
    SPEED_OPTIMIZED = 2


# ----------------------------------------------------------------------------------------
# coordinate manager
# ----------------------------------------------------------------------------------------
_SPLIT_PYRAMID = True
# a host-read-free pyramid queued in three lanes ordered by who waits for what (ops.build_pyramid_lanes); 0: one chain
_PYRAMID_LANES = __import__("os").environ.get("LIDIFF_PYRAMID_LANES", "1") != "0"


class CoordinateMap:
    """coords [M, 4] of one tensor stride + its hash table.  Host-read-free managers hand the rows over at their BOUND:
    `count` is then the int32 [1] device tensor holding the number of valid rows and `hint` the row count the host believes
    (the same level of an earlier pyramid of the same role) -- every size-dependent DECISION uses rows(), never coords.shape."""
    __slots__ = ("coords", "table", "ts", "count", "hint")

    def __init__(self, coords, table, ts, count=None, hint=None):
        self.coords, self.table, self.ts, self.count, self.hint = coords, table, ts, count, hint

    def rows(self) -> int:
        return self.coords.shape[0] if self.count is None else self.hint


class CoordinateManager:
    """Per-field cache of coordinate maps (keyed by tensor stride) and kernel maps (keyed by
    (ts_in, ts_out, kernel_size, transposed)) -- the reuse LiDiff relies on: across layers,
    across the cond/uncond forwards of a step (pipeline:149-151), and transposed convs landing
    on the encoder's maps so ME.cat is legal (SURVEY.md 8b 'Ownership / lifetime')."""

    def __init__(self, device):
        self.device = device
        self.maps: dict[int, CoordinateMap] = {}
        self.parents: dict[int, torch.Tensor] = {}      # ts_out -> parent row of every finer row
        self.kmaps: dict[tuple, torch.Tensor] = {}
        self.aux: dict = {}                             # derived per-map caches (match indices, ...)
        self.status = torch.zeros(1, dtype=torch.int32, device=device)
        self._async = None                              # (build stream, second build stream) -- set_async()
        self._ready = None                              # event: the field's points exist (recorded on the consumer's stream)
        self._on_level = None                           # callable(ts): queued on the second stream when level ts exists
        self._on_level_dev = None                       # callable(ts, rows_bound, d_count) -> [(aux key, owner, tensor, event)]:
        #                                                 the same, but inside the pyramid chain, before its sizes reach the host
        # True: insert() builds the whole pyramid (ops.build_pyramid: voxel map, the four strided maps, the kernel_size-3
        # maps and tail-map counts of the first two levels) with ONE host read instead of one per map.  Same maps.
        self.pyramid = False
        # Host-read-free building (DiffCompletion.read_free): `feed` (ops.SizeFeed) collects the sizes of the
        # pyramids of this field's role; with read_free the pyramid is built WITHOUT a host read -- maps at their bound, counts on
        # the device, kernel choices from `hints` (the sizes of the role's previous pyramid).  Fused inference plan only.
        self.feed = None
        self.read_free = False
        self.hints = None               # [rows of levels 0 .. 4, tail pairs of levels 0, 1] believed by the host
        self.exact_rows = None          # strides whose rows must come out EXACTLY as hinted (the condition's latent), checked later
        self.hint_lag = False           # exact sizes, but kernel choices from `hints` (the A/B twin of read_free: same choices)
        self.lane_up_orders = True      # a pyramid built in lanes also builds the decoder's offset-grouped pair lists (up_order)

    def rows(self, ts: int) -> int:
        """Row count of the map of stride ts as the host knows it (exact, or the hint of a read-free map); 0 if absent.
        With hint_lag (exact maps, decisions from an earlier pyramid's sizes) the hint wins wherever it exists."""
        m = self.maps.get(ts)
        if m is None:
            return 0
        if self.hint_lag and self.hints is not None and m.count is None:
            lv = ts.bit_length() - 1
            if lv < len(self.hints):
                return self.hints[lv]
        return m.rows()

    def count(self, ts: int):
        """int32 [1] device tensor with the valid rows of a read-free map, None for a map of exact size."""
        return self.maps[ts].count

    def tail_rows(self, ts: int):
        """Believed pair count of the tail map of stride ts (profiler bookkeeping only)."""
        tm = self.aux.get(("tail", ts))
        if tm is None or not getattr(tm, "bounded", False):
            return None if tm is None else tm.n
        lv, levels = ts.bit_length() - 1, int(math.log2(self.MAX_STRIDE))
        if self.hints is not None and levels + 1 + lv < len(self.hints):
            return self.hints[levels + 1 + lv]
        return None

    # -- asynchronous, on-demand building (DiffCompletion, round 3) ------------------------------------------------------
    def set_async(self, side, side2, ready=None, on_level=None, on_level_dev=None, side3=None):
        """Build every map of this manager ON DEMAND on `side` (the map-size reads synchronise that stream only) while the
        consumer's stream keeps running what is already queued; consumers join through lidiff_amd._lib.call().  on_level(ts)
        runs on `side2` as soon as the coordinate map of stride ts exists (the part -> full match of that level)."""
        from .. import _lib
        _lib.register_build_stream(side)
        _lib.register_build_stream(side2)
        if side3 is not None:           # third lane of a host-read-free pyramid (ops.build_pyramid_lanes): the deeper levels
            _lib.register_build_stream(side3)
        self._async, self._ready, self._on_level, self._on_level_dev = (side, side2, side3), ready, on_level, on_level_dev
        if ready is not None:
            side.wait_event(ready)
            side2.wait_event(ready)

    def clear_async(self):
        self._async = self._on_level = self._on_level_dev = None

    @contextlib.contextmanager
    def building(self, second: bool = False):
        """Context of every map-building call: the build stream when set_async() is on, the current stream otherwise."""
        if self._async is None:
            yield
            return
        from .. import _lib
        st = self._async[1 if second else 0]
        with torch.cuda.stream(st):
            yield
        if not second:                   # (the second stream's callbacks publish their results with their own events: a
            _lib.mark_pending(st)        #  consumer of level 0's match must not wait for the matches of all five levels)

    def _level_built(self, ts: int):
        if self._async is not None and self._on_level is not None:
            side, side2 = self._async[:2]
            ev = torch.cuda.Event()
            ev.record(side)
            side2.wait_event(ev)
            with self.building(second=True):
                self._on_level(ts)

    def _acquire(self, ts: int, up: bool = False):
        """A pyramid built in lanes (ops.build_pyramid_lanes) hands its maps over level by level: the first consumer of a level
        on another stream waits for THAT level's event -- the stem for level 0 only (~0.2 ms behind the points), not for the
        whole chain; `up`: the decoder's transposed maps, built last."""
        lane = getattr(self, "_lane", None)
        if lane is None:
            return
        from .. import _lib
        cur = torch.cuda.current_stream(self.device)
        if cur in _lib._BUILD_STREAMS:
            return                       # (builders are ordered by the lanes themselves)
        for key, ev in ((1, lane["ready"][1]), (ts, lane["ready"].get(ts)), ("up", lane["ready_up"] if up else None)):
            if ev is not None and (key, cur) not in lane["seen"]:
                cur.wait_event(ev)
                lane["seen"].add((key, cur))

    def insert(self, coords_i32: torch.Tensor, feats: torch.Tensor | None = None):
        """feats: the field's features, for the voxel mean to be queued INSIDE the pyramid chain where that pays (kept in
        self._feats0 for TensorField.sparse())."""
        if self.pyramid:
            return self._insert_pyramid(coords_i32, feats)
        with self.building():
            uniq, inverse, first_idx, table = ops.vox_unique(coords_i32, self.status)
        self.maps[1] = CoordinateMap(uniq, table, 1)
        self._level_built(1)
        return inverse, first_idx

    def _insert_pyramid(self, coords_i32: torch.Tensor, feats=None):
        levels = int(math.log2(self.MAX_STRIDE))
        second = self._async[1] if self._async is not None and _SPLIT_PYRAMID else None
        hook, early = None, []
        if second is not None and self._on_level_dev is not None:
            # per-level work that needs coordinates but no host-side size (the part -> full matches): queued INSIDE the chain,
            # with the row counts on the device, instead of behind the pyramid's host read
            def hook(lv, rows, d_count):
                early.append((lv, self._on_level_dev(1 << lv, rows, d_count)))
        free = False
        if self.feed is not None:
            if self.hints is None and self.feed.has_records():
                self.hints = self.feed.get()[1]      # the role's previous pyramid (waits for the device if it is not there yet)
            free = self.read_free and self.hints is not None
        third = self._async[2] if self._async is not None and len(self._async) > 2 else None
        if free and second is not None and third is not None and _PYRAMID_LANES:
            from .. import _lib
            h = self.hints
            # (a level's tail map is filled inside the chain when the layers will use it: the centre + tail rule of conv_bn_act;
            #  the decoder's pair lists for the maps large enough to take them: up_order())
            sparse = tuple(h[lv] >= 1024 and h[lv + 1] >= 0.85 * h[lv] for lv in (0, 1))
            ups = tuple(lv for lv in range(1, levels + 1) if self.lane_up_orders and h[lv - 1] >= self.UP_ORDER_MIN_ROWS)
            with self.building():
                sorted_lv = tuple(lv for lv in range(levels + 1)
                                  if ops.SPLIT3_SORTED and ops.SPLIT3_PRESORT and ops.split3_layer(1 << lv, h[lv], 2, 128, 0, 128))
                pyr = ops.build_pyramid_lanes(coords_i32, self.status, self.feed, second, third, strides=levels, on_level_dev=hook,
                                              feats=feats, tail_levels=sparse, up_pairs_levels=ups, sorted_levels=sorted_lv)
            # every map of the networks exists now and is handed over through per-level events (_acquire): no blanket join
            _lib.unmark_pending(self._async[0])
            self._feats0 = pyr.feats0
            self._lane = {"ready": pyr.ready, "ready_up": pyr.ready_up, "seen": set()}
            for lv in range(levels + 1):
                self.kmaps[(1 << lv, 1 << lv, 3, False)] = pyr.nbr3[lv]
            for lv in range(1, levels + 1):
                fine, coarse = 1 << (lv - 1), 1 << lv
                self.kmaps[(fine, coarse, 2, False)] = pyr.down[lv]
                self.kmaps[(coarse, fine, 2, True)] = pyr.up[lv]
                hit = pyr.up_lists.get(lv)
                self.aux[("up_order", coarse, fine)] = None if hit is None else (hit[3], hit[1])
                if hit is not None:
                    self.aux[("up_pairs", coarse, fine)] = hit[:3]
        else:
            with self.building():
                pyr = ops.build_pyramid(coords_i32, self.status, strides=levels, tail_levels=2, second_stream=second,
                                        on_level_dev=hook, feed=self.feed, read_free=free)
        self._counts = pyr.counts
        if free:           # rows the host takes as exact (exact_view) are checked when the device's counts arrive.  (A tail map above
            # its pair bound is flagged by the kernel that FILLS it -- only a map some layer really uses matters -- in the loop's
            # shared status word, which travels with every later record and is read at the end of the loop.)
            self.feed.expect(self.feed.seq, {ts.bit_length() - 1: self.hints[ts.bit_length() - 1] for ts in (self.exact_rows or ())})
        for lv, results in early:
            for key, owner, bound_tensor, event in results:
                self.aux[key] = (owner, bound_tensor if free else bound_tensor[:pyr.coords[lv].shape[0]], event)
        for lv in range(levels + 1):
            ts = 1 << lv
            if free:
                self.maps[ts] = CoordinateMap(pyr.coords[lv], pyr.tables[lv], ts, count=pyr.counts[lv:lv + 1], hint=self.hints[lv])
            else:
                self.maps[ts] = CoordinateMap(pyr.coords[lv], pyr.tables[lv], ts)
            if lv:
                self.parents[ts] = pyr.parents[lv]
        for lv, (nbr, tail) in enumerate(zip(pyr.nbr3, pyr.tails)):
            self.kmaps[(1 << lv, 1 << lv, 3, False)] = nbr
            self.aux[("tail", 1 << lv)] = tail
        # the per-level callbacks (DiffCompletion: the part -> full matches, 50-250 us kernels that fill the chip) are queued by
        # flush_levels() -- TensorField.sparse() calls it behind the voxel mean, so that the first convolution's inputs are
        # not stuck behind them
        self._deferred_levels = [] if hook is not None else [1 << lv for lv in range(levels + 1)]
        return pyr.inverse, pyr.first_idx

    def flush_levels(self):
        for ts in getattr(self, "_deferred_levels", ()):
            self._level_built(ts)
        self._deferred_levels = []

    def stride(self, ts: int, s: int) -> int:
        ts_out = ts * s
        if ts_out not in self.maps:
            with self.building():
                coarse, parent, table = ops.map_stride(self.maps[ts].coords, ts_out, self.status)
            self.maps[ts_out] = CoordinateMap(coarse, table, ts_out)
            self.parents[ts_out] = parent
            self._level_built(ts_out)
        return ts_out

    def kernel_map(self, ts_in: int, ts_out: int, ks: int, transposed: bool = False) -> torch.Tensor:
        self._acquire(max(ts_in, ts_out), up=transposed)
        key = (ts_in, ts_out, ks, transposed)
        nbr = self.kmaps.get(key)
        if nbr is None:
            with self.building():
                nbr = self._build_kernel_map(ts_in, ts_out, ks, transposed)
            self.kmaps[key] = nbr
        return nbr

    def _build_kernel_map(self, ts_in, ts_out, ks, transposed):
        if self.maps[ts_out].count is not None:      # read-free maps: tables at the bound, the row counts read on the device
            if transposed and ks == 2 and ts_in == 2 * ts_out and ts_in in self.parents:
                return ops.kernel_map_up_dev(self.maps[ts_out].coords, self.parents[ts_in], self.maps[ts_out].count, ts_out)
            if not transposed and ks == 2 and ts_out == 2 * ts_in and ts_out in self.parents:
                return ops.kernel_map_down_dev(self.maps[ts_in].coords, self.parents[ts_out], self.maps[ts_in].count, ts_in,
                                               self.maps[ts_out].coords.shape[0])
            if not transposed and ks == 3 and ts_in == ts_out:
                return ops.kernel_map_self_dev(self.maps[ts_out].coords, self.maps[ts_out].count, self.maps[ts_in].table, ts_in)
            raise RuntimeError("a host-read-free coordinate manager serves LiDiff's maps only (k3 s1, k2 s2 down / up)")
        if transposed:      # input = coarse map (ts_in), output = existing fine map (ts_out)
            if ks != 2 or ts_in != 2 * ts_out or ts_in not in self.parents:
                raise RuntimeError("transposed convolution supported for kernel_size=2, stride=2 "
                                   "onto a map created by the matching strided convolution")
            return ops.kernel_map_up(self.maps[ts_out].coords, self.parents[ts_in], ts_out)
        if ks == 2 and ts_out == 2 * ts_in and ts_out in self.parents:
            # the strided convolution's map straight from the parent array of the stride map (no lookups)
            return ops.kernel_map_down(self.maps[ts_in].coords, self.parents[ts_out], ts_in,
                                       self.maps[ts_out].coords.shape[0])
        return ops.kernel_map(self.maps[ts_out].coords, self.maps[ts_in].table, ks, ts_in,
                              self_map=(ts_in == ts_out and ks == 3))

    def tail_map(self, ts: int):
        """ops.TailMap of the kernel_size-3 map on stride ts (non-centre pairs by offset + CSR by output row); cached."""
        key = ("tail", ts)
        self._acquire(ts)
        if key not in self.aux:
            nbr = self.kernel_map(ts, ts, 3)
            with self.building():
                if self.maps[ts].count is not None:
                    self.aux[key] = ops.TailMap.on_device(nbr, self.maps[ts].count, self.status).fill()
                else:
                    self.aux[key] = ops.TailMap(nbr)
        elif getattr(self.aux[key], "_pending", None) is not None:      # counted in build_pyramid(): fill, no host read
            with self.building():
                self.aux[key].fill()
        return self.aux[key]

    def kernel_map_mask_sorted(self, ts: int):
        """(the kernel_size-3 table of level ts with its columns sorted by the rows' neighbour sets, the order): ops.mask_sorted_map,
        cached on the table (a pyramid built in lanes sorts the levels the split-operand kernel will take on its own stream)."""
        return ops.mask_sorted_map(self.kernel_map(ts, ts, 3))

    ORDER_MIN_ROWS = 30000      # smaller maps fit the L2 anyway

    def tile_order(self, ts: int):
        """Morton order of the map's rows for the sparse convolution's tiles (None for small maps); cached."""
        key = ("order", ts)
        if key not in self.aux:
            c = self.maps[ts].coords
            self.aux[key] = ops.tile_order(c, ts) if c.shape[0] >= self.ORDER_MIN_ROWS else None
        return self.aux[key]

    def kernel_map_ordered(self, ts_in: int, ts_out: int, ks: int, transposed: bool = False):
        """(neighbour table with its columns in tile order, the order) -- or (plain table, None)."""
        order = self.tile_order(ts_out)
        nbr = self.kernel_map(ts_in, ts_out, ks, transposed)
        if order is None:
            return nbr, None
        key = (ts_in, ts_out, ks, transposed, "ordered")
        hit = self.kmaps.get(key)
        if hit is None:
            hit = nbr.index_select(1, order.long()).contiguous()
            self.kmaps[key] = hit
        return hit, order

    UP_ORDER_MIN_ROWS = 4096
    MAX_STRIDE = 16             # the coarsest level of LiDiff's networks (four stride-2 stages)

    def up_order(self, ts_in: int, ts_out: int):
        """(neighbour table with its columns grouped by kernel offset, that row order) of the transposed kernel_size-2 /
        stride-2 map ts_in -> ts_out, or None for small maps.  Every output row of such a map has exactly ONE pair (its
        parent voxel, under the offset its position inside the parent cell selects), so in plain row order a 128-row tile
        holds ~16 pairs of each of the 8 offsets: eight almost empty MFMA stages per channel slab.  Grouped by offset a
        tile is 128 pairs of ONE offset -- a dense stage.  Results do not depend on the order (lidiff_spconv_fwd row_order)."""
        key = ("up_order", ts_in, ts_out)
        self._acquire(ts_in, up=True)
        if key not in self.aux:
            nbr = self.kernel_map(ts_in, ts_out, 2, True)
            if self.rows(ts_out) < self.UP_ORDER_MIN_ROWS:
                self.aux[key] = None
            else:
                # the ME-layout rulebook of the map lists its pairs by offset, then by output row: its output-row column IS the
                # order (one pair per row => a permutation), built by three small kernels without a host read
                with self.building():
                    pin, order, off = ops.rulebook_compact(nbr, total=nbr.shape[1], bounded=self.maps[ts_out].count is not None)
                    self.aux[key] = (nbr.index_select(1, order.long()).contiguous(), order)
                    self.aux[("up_pairs", ts_in, ts_out)] = (pin, order, off)
        return self.aux[key]

    def up_pairs(self, ts_in: int, ts_out: int):
        """The same map as a pair list grouped by offset -- (input row, output row, offset_ptr [9]) -- for
        ops.spconv_fwd_pairs (one pair per output row: the map IS its rulebook), or None for small maps."""
        if self.up_order(ts_in, ts_out) is None:
            return None
        return self.aux[("up_pairs", ts_in, ts_out)]

    def is_sparse_map(self, ts_in: int, ts_out: int, ks: int, transposed: bool = False, c_out: int = 128) -> bool:
        """Performance hint for lidiff_spconv_fwd (LIDIFF_CONV_SPARSE_MAP), from voxel counts the host already
        holds (no device sync): does the kernel map bring only a few pairs per offset and 128-row tile?
        r = M(2 ts) / M(ts) close to 1 means the voxels of stride ts are isolated (each coarse voxel holds ~1 of
        them), i.e. few neighbours.  Results do not depend on the hint.
        The hint selects the kernel family whose tiles can pack several offsets into one stage.  Measured on the 180k-point
        scan (profiles/r03_hint_sweep.txt): for the 128-column tiles it pays up to ~2 neighbours per voxel (r >= 0.85); the
        narrow tiles (C_out 96 / 64: 3 x 2 and 4 x 2 wave grids, eight offsets per stage) win up to ~7 neighbours per voxel --
        +38 % on 64 -> 64 at stride 4 (4.3 neighbours, r = 0.69), +36 % on 96 -> 96 at stride 2 with 3.6 (r = 0.73), equal at
        11 (r = 0.38) -- so they take it for r >= 0.5."""
        m = self.rows
        if ks == 1:
            return False
        if ks == 2:                                   # stride-2 down: 16 / r pairs per offset and tile; up: 16
            if transposed:
                return True
            return m(ts_out) >= 0.67 * max(1, m(ts_in))
        if self._async is not None and 2 * ts_in not in self.maps and ts_in in self.maps and ts_in < self.MAX_STRIDE:
            self.stride(ts_in, 2)                     # on-demand building: the hint of a level needs the next level's size
        coarse = m(2 * ts_in)
        return coarse > 0 and coarse >= (0.85 if c_out % 128 == 0 else 0.5) * m(ts_in)

    def prebuild_strides(self, max_stride: int = 16):
        """The strided coordinate maps alone (levels 2 .. max_stride): everything that needs only coordinates -- e.g. the
        part -> full matches of MinkUNetDiff -- can start as soon as these exist."""
        ts = 1
        while ts < max_stride:
            ts = self.stride(ts, 2)

    def prebuild(self, max_stride: int = 16, tail_maps: bool = True, up_orders: bool = False):
        """Every map the networks will ask for, built now (MinkGlobalEnc / MinkUNetDiff / MinkUNet: four stride-2 levels, a
        kernel_size-3 map per level, the kernel_size-2 maps down and back up, the tail maps of the low-density levels).
        The builders read map sizes back to the host; doing all of it in one place lets DiffCompletion run it on a side
        stream, under the convolutions of another tensor, instead of level by level in the middle of a network."""
        ts = 1
        while True:
            self.kernel_map(ts, ts, 3)
            if ts == max_stride:
                break
            nxt = self.stride(ts, 2)
            self.kernel_map(ts, nxt, 2)
            self.kernel_map(nxt, ts, 2, True)
            if up_orders:
                self.up_order(nxt, ts)
            ts = nxt
        for ts in list(self.maps):                      # the sparse-map hint of a level needs the next level's size
            if tail_maps and self.rows(ts) >= 1024 and self.is_sparse_map(ts, ts, 3):
                self.tail_map(ts)

    def prebuild_rulebooks(self):
        """The ME-layout rulebooks of every kernel map built so far (the weight-gradient kernel walks them), with one host
        read for all of them (ops.build_rulebooks) instead of one per map in the middle of the backward pass."""
        ops.build_rulebooks([t for t in self.kmaps.values() if isinstance(t, torch.Tensor) and t.dim() == 2])

    def tensors(self):
        """Every device tensor this manager holds (for record_stream when it was built on another stream)."""
        out = [self.status]
        if getattr(self, "_counts", None) is not None:
            out.append(self._counts)
        for m in self.maps.values():
            out += [m.coords, m.table.keys, m.table.vals]
        out += list(self.parents.values())
        out += [t for t in self.kmaps.values() if isinstance(t, torch.Tensor)]
        for t in self.kmaps.values():          # (the mask-sorted twin of a table and its row order hang on the table: ops.mask_sorted_map)
            if isinstance(t, torch.Tensor) and getattr(t, "_lidiff_mask_sorted", None) is not None:
                out += list(t._lidiff_mask_sorted)
        for v in self.aux.values():
            for t in (v if isinstance(v, (tuple, list)) else [v]):
                if isinstance(t, torch.Tensor):
                    out.append(t)
                elif isinstance(t, ops.TailMap):
                    out += [q for q in (t.ptr, t.nbr, t.idx, t.pair_in, t.off) if q is not None]
                    out += [q for q in (getattr(t, "_pending", None) or ()) if isinstance(q, torch.Tensor)]
        return out

    def record_stream(self, stream):
        for t in self.tensors():
            t.record_stream(stream)

    def check(self):
        """Raise if a kernel flagged a coordinate outside the hash-key range (host sync)."""
        s = int(self.status.item())
        if s:
            self.status.zero_()          # (raised once: the word may be shared with the fields of a later loop)
        if s & ops.STATUS_KEY_RANGE:
            raise RuntimeError("coordinate outside [-32768, 32767]: not representable in the 64-bit key")
        if s & ops.STATUS_HASH_FULL:
            raise RuntimeError("coordinate hash table overflow")
        if s & ops.STATUS_BOUND:
            raise RuntimeError("a device-side count exceeded the bound of a host-read-free step (the step must be redone)")


# ----------------------------------------------------------------------------------------
# autograd functions
# ----------------------------------------------------------------------------------------
class _VoxelMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, inverse, m):
        out, counts = ops.vox_mean(feats, inverse, m)
        ctx.save_for_backward(inverse, counts)
        return out

    @staticmethod
    def backward(ctx, g):
        inverse, counts = ctx.saved_tensors
        return ops.vox_mean_bwd(g, inverse, counts), None, None


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        ctx.save_for_backward(idx)
        ctx.m = src.shape[0]
        return ops.gather_rows(src, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ops.scatter_add_rows(g, idx, ctx.m), None


class _GatherMulRows(torch.autograd.Function):
    """x * table[idx] (the conditioning multiply with its MLP evaluated on a table, minkunet.py:431) as ONE pass forward and one
    for the gradient of x -- no [M, C] copy of the gathered weights; the table's gradient is the segment sum of g * x."""

    @staticmethod
    def forward(ctx, x, table, idx):
        ctx.save_for_backward(x, table, idx)
        return ops.gather_mul_rows(x, table, idx)

    @staticmethod
    def backward(ctx, g):
        x, table, idx = ctx.saved_tensors
        g = g.contiguous()
        gx = ops.gather_mul_rows(g, table, idx) if ctx.needs_input_grad[0] else None
        gt = ops.scatter_add_rows(g * x, idx, table.shape[0]) if ctx.needs_input_grad[1] else None
        return gx, gt, None


class _SparseConv(torch.autograd.Function):
    """out = sum_k in[nbr[k]] @ W[k].  backward: dX is the same operator over the swapped map
    with W^T (for a centred odd kernel on one map the swap is k -> K-1-k); dW = gather^T @ g."""

    @staticmethod
    def forward(ctx, x, kernel, nbr, nbr_swapped, m_out, flip, sparse_map=False):
        ctx.flip = flip
        w3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        ctx.bf16 = ops.TRAIN_OPERANDS == "bf16" and ops.bf16_conv_applies(w3.shape[1], 0, w3.shape[2], sparse_map)
        # the weight-gradient kernel does not care about the map's density: bf16 operands on every eligible layer
        ctx.bf16_dw = ops.TRAIN_OPERANDS == "bf16" and w3.shape[1] % 32 == 0 and w3.shape[2] % 32 == 0
        # bf16 activations in HBM (ops.BF16_ROWS): the kernels gather the input's bf16 shadow -- one cast per tensor, shared by
        # this forward, the weight gradient (which keeps ONLY the shadow: half the saved bytes) and the other layers on x
        ctx.rows16 = ops.BF16_ROWS and (ctx.bf16 or ctx.bf16_dw) and x.is_contiguous() and x.dtype == torch.float32
        x16 = ops.cast_bf16(x) if ctx.rows16 else None
        ctx.m_in = x.shape[0]
        ctx.save_for_backward(x16 if ctx.rows16 and ctx.bf16_dw else x, kernel, nbr, nbr_swapped)
        if ctx.bf16:
            return ops.spconv_fwd_bf16(x16 if ctx.rows16 else x, kernel, nbr, m_out)
        return ops.spconv_fwd(x, kernel, nbr, m_out)

    @staticmethod
    def backward(ctx, g):
        x, kernel, nbr, nbr_swapped = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = None
        w3 = kernel if kernel.dim() == 3 else kernel.unsqueeze(0)
        need_w16 = ctx.needs_input_grad[1] and ctx.bf16_dw and x.dtype == torch.bfloat16
        g16 = ops.cast_bf16(g) if ctx.rows16 and ((ctx.needs_input_grad[0] and ctx.bf16) or need_w16) else None
        if ctx.needs_input_grad[0] and ctx.bf16:
            gx = ops.spconv_fwd_bf16(g if g16 is None else g16, kernel, nbr_swapped, ctx.m_in, transposed=True, flip=ctx.flip)
        elif ctx.needs_input_grad[0]:
            wt = (w3.flip(0) if ctx.flip else w3).transpose(1, 2).contiguous()
            gx = ops.spconv_fwd(g, wt, nbr_swapped, ctx.m_in)
        if ctx.needs_input_grad[1]:
            gw = ops.spconv_bwd_w(x, g16 if need_w16 else g, nbr, w3.shape[0], bf16=ctx.bf16_dw).reshape(kernel.shape)
        return gx, gw, None, None, None, None, None


# ----------------------------------------------------------------------------------------
# tensors
# ----------------------------------------------------------------------------------------
class TensorField:
    """ME.TensorField (pipeline:74-80): per-point features + float coordinates [N, 1+D] whose
    column 0 is the batch index.  ``.sparse()`` floors the coordinates to int32, hashes them
    into unique voxels and averages member features (UNWEIGHTED_AVERAGE)."""

    def __init__(self, features, coordinates, quantization_mode=SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                 minkowski_algorithm=MinkowskiAlgorithm.DEFAULT, coordinate_manager=None, device=None):
        if quantization_mode not in (SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,):
            raise NotImplementedError("LiDiff only uses UNWEIGHTED_AVERAGE quantisation")
        if device is not None:
            features, coordinates = features.to(device), coordinates.to(device)
        ops.require_device(features, coordinates)
        if coordinates.dim() != 2 or coordinates.shape[1] != 4:
            raise ValueError("coordinates must be [N, 4] = (batch, x, y, z)")
        if features.shape[0] != coordinates.shape[0]:
            raise ValueError("features and coordinates disagree on the number of points")
        self._F = features
        self._C = coordinates
        self.quantization_mode = quantization_mode
        self.coordinate_manager = coordinate_manager or CoordinateManager(features.device)
        self.inverse_mapping = None
        self._sparse = None          # the voxelised tensor, once built without autograd (DiffCompletion.prepare)
        self.prepared = None         # event: maps and voxel features ready (built on a side stream)
        self.ready = None            # event: features / coordinates written (recorded by whoever wants to hand the field over)

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self._C

    @property
    def device(self):
        return self._F.device

    def sparse(self) -> "SparseTensor":
        mgr = self.coordinate_manager
        no_graph = not (torch.is_grad_enabled() and self._F.requires_grad)
        if self.inverse_mapping is None:
            with mgr.building():
                ci = self._C if self._C.dtype == torch.int32 else ops.coords_floor(self._C)
            self.inverse_mapping, _ = mgr.insert(ci, feats=self._F if no_graph and self._F.dtype == torch.float32 else None)
        if self._sparse is not None:
            mgr.flush_levels()
            return self._sparse
        m = mgr.maps[1].coords.shape[0]
        f = getattr(mgr, "_feats0", None)        # the voxel mean was queued inside the pyramid chain (build_pyramid_lanes)
        mgr._feats0 = None
        if f is None or not no_graph:
            with mgr.building():
                f = _VoxelMean.apply(self._F.float(), self.inverse_mapping, m)
        mgr.flush_levels()              # the per-level callbacks (matches), behind the voxel mean
        sp = SparseTensor(f, tensor_stride=1, coordinate_manager=mgr)
        if not (torch.is_grad_enabled() and self._F.requires_grad):
            self._sparse = sp        # same features for every caller of this field (no graph attached)
        return sp


class SparseTensor:
    """ME.SparseTensor: features [M, C] on the coordinate map `tensor_stride` of a manager."""

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_manager=None, device=None):
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("need coordinates or a coordinate manager")
            # ME.SparseTensor(features, coordinates): quantise like a field and average duplicates
            field = TensorField(features, coordinates.float() if coordinates.is_floating_point() else coordinates,
                                device=device)
            sp = field.sparse()
            features, coordinate_manager, tensor_stride = sp.F, sp.coordinate_manager, 1
        self._F = features
        self.tensor_stride = int(tensor_stride)
        self.coordinate_manager = coordinate_manager
        self.replicas = 1          # > 1: that many feature matrices stacked row-wise on ONE coordinate map (CFG pair)
        self._rows = None          # exact_view(): the first _rows rows of a read-free map, taken as its exact size

    @property
    def F(self):
        return self._F

    @property
    def C(self):
        c = self.coordinate_manager.maps[self.tensor_stride].coords
        return c if self._rows is None else c[:self._rows]

    @property
    def device(self):
        return self._F.device

    def exact_view(self) -> "SparseTensor":
        """A tensor on a host-read-free map cut to the rows the host BELIEVES it has (the map's hint), to be used where the
        row count shapes host-side work (the condition's latent: MLP tables, match targets).  The manager registers the belief
        with its SizeFeed, which checks it against the device's count when that arrives.  Exact maps: self."""
        m = self.coordinate_manager.maps[self.tensor_stride]
        if m.count is None or self.replicas != 1:
            return self
        out = SparseTensor(self._F[:m.hint], tensor_stride=self.tensor_stride, coordinate_manager=self.coordinate_manager)
        out._rows = m.hint
        return out

    def _like(self, features):
        out = SparseTensor(features, tensor_stride=self.tensor_stride, coordinate_manager=self.coordinate_manager)
        out.replicas = self.replicas
        return out

    def replicate(self, n: int) -> "SparseTensor":
        """n copies of the features stacked row-wise on the same coordinate map (fused execution only)."""
        out = self._like(self._F.repeat(n, 1))
        out.replicas = self.replicas * n
        return out

    def __mul__(self, other):
        if isinstance(other, SparseTensor):
            _same_map(self, other)
            other = other.F
        return self._like(self._F * other)

    def __add__(self, other):
        if isinstance(other, SparseTensor):
            _same_map(self, other)
            other = other.F
        return self._like(self._F + other)

    def slice(self, field: TensorField) -> TensorField:
        """F_vox[inverse_mapping] -> per-point features (minkunet.py:497,619)."""
        if field.coordinate_manager is not self.coordinate_manager or self.tensor_stride != 1:
            raise RuntimeError("slice needs the stride-1 tensor of the field's own coordinate manager")
        out = TensorField.__new__(TensorField)
        out._F = _GatherRows.apply(self._F, field.inverse_mapping)
        out._C = field._C
        out.quantization_mode = field.quantization_mode
        out.coordinate_manager = field.coordinate_manager
        out.inverse_mapping = field.inverse_mapping
        out._sparse = out.prepared = out.ready = None
        return out


def _same_map(a: SparseTensor, b: SparseTensor):
    if a.coordinate_manager is not b.coordinate_manager or a.tensor_stride != b.tensor_stride:
        raise RuntimeError("sparse tensors live on different coordinate maps")


def cat(*tensors):
    """ME.cat: channel concat of tensors on the same coordinate map (minkunet.py:464,...)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    for t in tensors[1:]:
        _same_map(tensors[0], t)
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))


# ----------------------------------------------------------------------------------------
# modules (state-dict compatible with ME: `.kernel`, `.bn.*`)
# ----------------------------------------------------------------------------------------
class _ConvBase(nn.Module):
    transposed = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 dimension=None):
        super().__init__()
        if dimension != 3:
            raise NotImplementedError("LiDiff is 3-D: dimension must be 3")
        if dilation != 1:
            raise NotImplementedError("dilation != 1 is not used by LiDiff")
        if bias:
            raise NotImplementedError("LiDiff's convolutions have no bias")
        if (kernel_size, stride) not in ((3, 1), (2, 2), (1, 1)):
            raise NotImplementedError(f"kernel_size={kernel_size}, stride={stride} is not on LiDiff's path")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation, self.dimension = kernel_size, stride, dilation, dimension
        k_vol = kernel_size ** 3
        shape = (in_channels, out_channels) if k_vol == 1 else (k_vol, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        self.bias = None
        self.reset_parameters()

    def reset_parameters(self):
        # ME: U(-1/sqrt(n), 1/sqrt(n)), n = (C_out if transposed else C_in) * kernel_volume (App. A.6)
        n = (self.out_channels if self.transposed else self.in_channels) * self.kernel_size ** 3
        with torch.no_grad():
            self.kernel.uniform_(-1.0 / math.sqrt(n), 1.0 / math.sqrt(n))

    def maps(self, x: SparseTensor, swapped: bool = True):
        """(nbr, nbr_swapped, ts_out, flip) for input x; builds / reuses the manager's maps.  swapped=False: the forward-only
        caller does not need the swapped map (the input gradient's) -- None in its place."""
        mgr, ts = x.coordinate_manager, x.tensor_stride
        if self.kernel_size == 1:
            return None, None, ts, False
        if self.transposed:
            ts_out = ts // self.stride
            if ts_out < 1 or ts_out not in mgr.maps:
                raise RuntimeError("transposed convolution needs the finer map created by the encoder")
            return (mgr.kernel_map(ts, ts_out, self.kernel_size, True),
                    mgr.kernel_map(ts_out, ts, self.kernel_size, False) if swapped else None, ts_out, False)
        if self.stride == 1:
            nbr = mgr.kernel_map(ts, ts, self.kernel_size)
            return nbr, nbr, ts, True
        ts_out = mgr.stride(ts, self.stride)
        return (mgr.kernel_map(ts, ts_out, self.kernel_size, False),
                mgr.kernel_map(ts_out, ts, self.kernel_size, True) if swapped else None, ts_out, False)

    def sparse_hint(self, x: SparseTensor, ts_out: int) -> bool:
        return x.coordinate_manager.is_sparse_map(x.tensor_stride, ts_out, self.kernel_size, self.transposed,
                                                  self.out_channels)

    def forward(self, x: SparseTensor) -> SparseTensor:
        nbr, nbr_sw, ts_out, flip = self.maps(x)
        mgr = x.coordinate_manager
        if mgr.maps[ts_out].count is not None:
            raise RuntimeError("host-read-free coordinate maps serve the fused inference plan only")
        m_out = mgr.maps[ts_out].coords.shape[0]
        if torch.is_grad_enabled() and (x.F.requires_grad or self.kernel.requires_grad):
            f = _SparseConv.apply(x.F, self.kernel, nbr, nbr_sw, m_out, flip, self.sparse_hint(x, ts_out))
        else:
            f = ops.spconv_fwd(x.F, self.kernel, nbr, m_out, sparse_map=self.sparse_hint(x, ts_out))
        return SparseTensor(f, tensor_stride=ts_out, coordinate_manager=mgr)

    def extra_repr(self):
        return (f"in={self.in_channels}, out={self.out_channels}, kernel_size={self.kernel_size}, "
                f"stride={self.stride}, transposed={self.transposed}")


class MinkowskiConvolution(_ConvBase):
    """ME.MinkowskiConvolution (minkunet.py:17,53,61,72,94,97,...)."""


class MinkowskiConvolutionTranspose(_ConvBase):
    """ME.MinkowskiConvolutionTranspose(ks=2, stride=2) (minkunet.py:36): upsamples onto the
    encoder's existing finer map through the swapped fine->coarse kernel map."""
    transposed = True


class MinkowskiBatchNorm(nn.Module):
    """ME.MinkowskiBatchNorm = nn.BatchNorm1d on F (child module `.bn`, which LiDiff's
    weight_initialization relies on, minkunet.py:128-132)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x: SparseTensor) -> SparseTensor:
        bn = self.bn
        # training: batch statistics and the normalisation through the HIP kernels of norm.hip (one pass each, deterministic;
        # torch's channels-last batch-norm kernels run at a tenth of the HBM rate on these shapes); the sync variant
        # (ops.SyncBatchNorm1d) takes the same kernels around one all-reduce of the per-channel sums; momentum = None
        # (cumulative average) stays on torch
        if ops.bn_module_fused(bn) and ops.bn_fused_applies(bn, x.F):
            return x._like(ops.batch_norm_train(x.F, bn))
        return x._like(bn(x.F))


class MinkowskiSyncBatchNorm(MinkowskiBatchNorm):
    """ME.MinkowskiSyncBatchNorm: statistics all-reduced over the process group (RCCL).  The child `.bn` is
    ops.SyncBatchNorm1d -- an nn.BatchNorm1d subclass (same state-dict keys) on the norm.hip kernels, so the fused BN + ReLU and
    BN + shortcut + ReLU forms of the training path (minkunet.py) stay in place under data parallelism."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True,
                 process_group=None):
        nn.Module.__init__(self)
        self.bn = ops.SyncBatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                      track_running_stats=track_running_stats, process_group=process_group)

    @classmethod
    def convert_sync_batchnorm(cls, module, process_group=None):
        """train.py:90 / train_refine.py:58: swap every MinkowskiBatchNorm for the sync version."""
        out = module
        if isinstance(module, MinkowskiBatchNorm) and not isinstance(module, MinkowskiSyncBatchNorm):
            src = module.bn
            out = cls(src.num_features, src.eps, src.momentum, src.affine, src.track_running_stats, process_group)
            if src.affine:
                with torch.no_grad():
                    out.bn.weight = src.weight
                    out.bn.bias = src.bias
            out.bn.running_mean = src.running_mean
            out.bn.running_var = src.running_var
            out.bn.num_batches_tracked = src.num_batches_tracked
        for name, child in module.named_children():
            if out is module:
                new = cls.convert_sync_batchnorm(child, process_group)
                if new is not child:
                    module.add_module(name, new)
        return out


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x: SparseTensor) -> SparseTensor:
        return x._like(torch.relu(x.F))
