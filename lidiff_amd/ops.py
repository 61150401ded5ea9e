"""Tensor-level wrappers over the C ABI (include/lidiff_amd.h): allocate outputs with torch,
pass raw device pointers + the current HIP stream, return torch tensors.  GPU only.

Each function names the reference call site whose native kernel it stands in for (paths
relative to /root/reference/lidiff).
"""
from __future__ import annotations

import contextlib
import math
import os

import torch

from . import _lib
from ._lib import call, ptr, require_device, stream_ptr

STATUS_KEY_RANGE = 1
STATUS_HASH_FULL = 2
STATUS_F16_RANGE = 8      # a value beyond fp16's range met the two-piece fp16 split (spconv_fwd_split3 with SPLIT_PIECES = 2)
STATUS_BOUND = 4          # a device-side count exceeded the bound a buffer was sized for (host-read-free steps)


class HashTable:
    """Open-addressing table of one coordinate map: packed 64-bit key -> row id."""

    def __init__(self, n_rows: int, device, pool: "BytePool | None" = None):
        self.cap = _lib.load().lidiff_hash_capacity(int(n_rows))
        if pool is None:
            self.keys = torch.empty(self.cap, dtype=torch.int64, device=device)
            self.vals = torch.empty(self.cap, dtype=torch.int32, device=device)
        else:                        # (views into the pool once it is committed: BytePool.commit)
            self.keys, self.vals = pool.reserve((self.cap,), torch.int64), pool.reserve((self.cap,), torch.int32)
            pool.on_commit(self, "keys", "vals")


class BytePool:
    """Several tensors that all START as the same byte pattern, carved out of ONE allocation cleared by ONE fill launch: the
    hash tables and neighbour tables of a coordinate pyramid (0xFF: the empty key, row -1) and its zeroed workspaces.  Round 5
    queued a clear per table -- 90 per denoising step --; the library calls take `preinit = 1` for pooled tables.
    reserve() hands out a ticket, commit(byte) allocates + fills and turns the tickets into views (256-byte aligned)."""

    def __init__(self, device):
        self.device, self.items, self.total, self.buf, self.fix = device, [], 0, None, []

    def reserve(self, shape, dtype):
        nbytes = int(torch.empty((), dtype=dtype).element_size())
        for d in shape:
            nbytes *= int(d)
        self.items.append((self.total, nbytes, tuple(int(d) for d in shape), dtype))
        self.total += -(-max(nbytes, 1) // 256) * 256
        return len(self.items) - 1

    def on_commit(self, obj, *attrs):
        self.fix.append((obj, attrs))

    def commit(self, byte: int):
        self.buf = torch.empty(max(self.total, 256), dtype=torch.uint8, device=self.device)
        self.buf.fill_(byte)
        for obj, attrs in self.fix:
            for a in attrs:
                setattr(obj, a, self.view(getattr(obj, a)))
        return self

    def view(self, ticket: int) -> torch.Tensor:
        off, nbytes, shape, dtype = self.items[ticket]
        return self.buf[off:off + nbytes].view(dtype).view(shape)


def _workspace(n_rows: int, device):
    nbytes = _lib.load().lidiff_unique_workspace_bytes(int(n_rows))
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def coords_floor(coords_f: torch.Tensor) -> torch.Tensor:
    """TensorField.sparse() quantisation (pipeline:149): floor float coords -> int32 [N,4]."""
    require_device(coords_f)
    coords_f = coords_f.contiguous().float()
    out = torch.empty(coords_f.shape, dtype=torch.int32, device=coords_f.device)
    call("lidiff_coords_floor", ptr(coords_f), coords_f.shape[0], ptr(out), stream_ptr())
    return out


def vox_unique(coords: torch.Tensor, status: torch.Tensor):
    """Voxel hashing (pipeline:149, minkunet.py:135,597).  Returns (uniq[M,4] int32,
    inverse[N] int64, first_idx[M] int32, HashTable).  One device->host read of M."""
    require_device(coords)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    table = HashTable(n, dev)
    uniq = torch.empty((n, 4), dtype=torch.int32, device=dev)
    first_idx = torch.empty(n, dtype=torch.int32, device=dev)
    inverse = torch.empty(n, dtype=torch.int64, device=dev)
    d_m = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(n, dev)
    call("lidiff_vox_unique", ptr(coords), n, ptr(table.keys), ptr(table.vals), table.cap,
         ptr(uniq), ptr(first_idx), ptr(inverse), ptr(d_m), ptr(status), ptr(ws), 0, stream_ptr())
    m = int(d_m.item())
    return uniq[:m], inverse, first_idx[:m], table


def map_stride(coords: torch.Tensor, s_out: int, status: torch.Tensor):
    """Strided coordinate map (minkunet.py:13-29).  Returns (coarse[Mc,4], parent[M] int32,
    HashTable of the coarse map)."""
    require_device(coords)
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    table = HashTable(n, dev)
    coarse = torch.empty((n, 4), dtype=torch.int32, device=dev)
    parent = torch.empty(n, dtype=torch.int32, device=dev)
    d_m = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = _workspace(n, dev)
    call("lidiff_map_stride", ptr(coords), n, int(s_out), ptr(table.keys), ptr(table.vals), table.cap,
         ptr(coarse), ptr(parent), ptr(d_m), ptr(status), ptr(ws), stream_ptr())
    m = int(d_m.item())
    return coarse[:m], parent, table


PYRAMID_TRACE = None     # debugging: a list -> build_pyramid appends (label, host time) pairs (tools/step_timeline.py)


def _trace(label):
    if PYRAMID_TRACE is not None:
        import time
        ev = None
        if label in ("enter", "tails queued", "done"):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        PYRAMID_TRACE.append((label, time.perf_counter(), ev))


PINNED_READ = True
_PINNED: dict = {}


class Pyramid:
    """What build_pyramid() hands over: per level the coordinate rows, hash table and (levels >= 1) the parent array of the
    finer level; for the first `tail_levels` levels also the kernel_size-3 self map and the phase-1 state of its tail map."""
    __slots__ = ("coords", "tables", "parents", "inverse", "first_idx", "nbr3", "tails", "counts", "bound", "fast", "feats0",
                 "ready", "ready_up", "down", "up", "up_lists")


# pair bound of a tail map built without a host read, per row of the map's bound: centre + tail is only chosen for maps with at
# most ~2 non-centre neighbours per voxel (CoordinateManager.is_sparse_map), so 3 per POINT is generous; a map that exceeds it
# raises STATUS_BOUND on the device and the step is redone with exact sizes
TAIL_PAIR_BOUND = 3


class SizeFeed:
    """Host-visible sizes of the coordinate pyramids of one ROLE in a denoising loop (x_t / condition / unconditional), one
    record per pyramid in build order.  A pyramid built with a host read pushes its sizes directly; a pyramid built WITHOUT one
    (build_pyramid(read_free=True)) has the device write them into pinned, device-mapped memory (lidiff_publish_words) -- no
    device->host copy, no synchronisation: the host polls the record's sequence number when it next needs the sizes, which is
    one step later (kernel-choice hints and the validation of the bounds it sized buffers by)."""
    SLOTS = 8
    WORDS = 64

    def __init__(self, device):
        self.device = device
        self.ring = torch.zeros((self.SLOTS, self.WORDS), dtype=torch.int32).pin_memory()
        self.view = self.ring.numpy()
        # the device writes the ring itself when the pinned block is mapped into its address space (the normal case:
        # hipHostMalloc); otherwise two small asynchronous copies per record (the sizes, then the sequence number)
        import ctypes
        dptr = ctypes.c_void_p()
        rc = _lib.load().lidiff_host_device_pointer(ctypes.c_void_p(self.ring.data_ptr()), ctypes.byref(dptr))
        self.dev_base = dptr.value if rc == 0 and dptr.value else None
        self.seq = 0                 # records announced so far
        self.seq_base = 0            # records before this one belong to an earlier scan (reset())
        self.done = {}               # seq -> (status, [sizes]) read back or pushed
        self.n_words = {}
        self.slot_owner = {}         # ring slot -> the device-written record it holds (host-pushed records use no slot)
        self.checks = {}             # seq -> {level: rows the host took as exact}
        self.bad = None              # first failed validation (message), sticky until reset()

    def reset(self):
        """Forget every record (a new scan: its first pyramid is built with a host read again).  Records the device still owes
        are read if the feed is healthy; after a failure (a record that never arrived, a voided loop) they are DISCARDED -- a
        publish launch that never ran must not make every later loop wait for it."""
        if self.bad is None:
            self.drain(timeout_s=5.0)
        self._discard_pending()
        self.done.clear()
        self.checks.clear()
        self.seq_base = self.seq
        self.bad = None

    def _discard_pending(self):
        for seq in list(self.n_words):
            del self.n_words[seq]
            self.checks.pop(seq, None)
        self.slot_owner = {s: q for s, q in self.slot_owner.items() if q in self.n_words}

    def has_records(self) -> bool:
        return self.seq > self.seq_base

    def expect(self, seq: int, exact_rows: dict, maxima: dict | None = None):
        """What the host assumed about record `seq`: words that must equal / must not exceed a value."""
        self.checks[seq] = (dict(exact_rows), dict(maxima or {}))

    def drain(self, timeout_s: float = 30.0):
        """Read every record the device still owes (end of a loop: all checks done).  Returns self.bad."""
        for seq in sorted(self.n_words):
            if seq in self.n_words:
                self.get(seq, timeout_s)
        return self.bad

    def _check(self, seq, status, sizes):
        exact, maxima = self.checks.pop(seq, ({}, {}))
        if self.bad is not None:
            return
        if status & STATUS_BOUND:
            self.bad = f"record {seq}: a device-side count exceeded its bound (status {status})"
        for lv, rows in exact.items():
            if sizes[lv] != rows:
                self.bad = f"record {seq}: level {lv} has {sizes[lv]} rows, the host took {rows} as exact"
        for i, top in maxima.items():
            if sizes[i] > top:
                self.bad = f"record {seq}: word {i} = {sizes[i]} exceeds the bound {top} the host sized a buffer by"

    def push_host(self, sizes, status: int = 0):
        self.seq += 1
        self.done[self.seq] = (int(status), [int(v) for v in sizes])
        self._trim()
        return self.seq

    def publish(self, counts: torch.Tensor, status: torch.Tensor | None):
        """Queue the device-side write of `counts` (int32 [k]) and *status into this feed's next record."""
        self.seq += 1
        slot = self.ring[self.seq % self.SLOTS]
        prev = self.slot_owner.get(self.seq % self.SLOTS)
        if prev is not None and prev in self.n_words:
            self.get(prev)                             # the slot's previous record has not been consumed yet: do it now
        self.slot_owner[self.seq % self.SLOTS] = self.seq
        self.n_words[self.seq] = counts.numel()
        if self.dev_base is not None:
            call("lidiff_publish_words", ptr(counts), counts.numel(), ptr(status), self.dev_base + 4 * self.WORDS * (self.seq % self.SLOTS),
                 self.seq, stream_ptr())
        else:
            body = torch.cat([status.reshape(1) if status is not None else counts.new_zeros(1), counts])
            slot[1:1 + body.numel()].copy_(body, non_blocking=True)
            slot[0:1].copy_(torch.full((1,), self.seq, dtype=torch.int32, device=counts.device), non_blocking=True)
        return self.seq

    def get(self, seq: int | None = None, timeout_s: float = 30.0):
        """(status, sizes) of record `seq` (default: the latest); waits for the device if it has not written it yet."""
        seq = self.seq if seq is None else seq
        hit = self.done.get(seq)
        if hit is not None:
            return hit
        if seq not in self.n_words:
            raise KeyError(f"size record {seq} is gone")
        import time
        row = self.view[seq % self.SLOTS]
        t0 = time.perf_counter()
        while int(row[0]) != seq:                      # the device has not got there yet (plain memory reads, no HIP call)
            if time.perf_counter() - t0 > timeout_s:
                # a publish launch that never ran (launch error, an exception inside a capture) or a device lagging far behind
                # (a profiler, a shared GPU): the feed is marked bad -- the loop is redone with exact sizes -- and every record
                # still owed is dropped, so that nothing waits for this one again
                self.bad = self.bad or "record %d was never published by the device (waited %.0f s)" % (seq, timeout_s)
                self._discard_pending()
                older = [k for k in self.done if k < seq]
                if not older:
                    raise RuntimeError("SizeFeed: " + self.bad)
                return (STATUS_BOUND, list(self.done[max(older)][1]))     # stale sizes as hints: the loop is void anyway
            time.sleep(0)
        n = self.n_words.pop(seq)
        hit = self.done[seq] = (int(row[1]), [int(v) for v in row[2:2 + n]])
        self._check(seq, *hit)
        self._trim()
        return hit

    def _trim(self):
        for k in [k for k in self.done if k <= self.seq - 2 * self.SLOTS]:
            del self.done[k]


def build_pyramid(coords: torch.Tensor, status: torch.Tensor, strides: int = 4, tail_levels: int = 2,
                  second_stream=None, on_level_dev=None, feed: "SizeFeed | None" = None, read_free: bool = False) -> Pyramid:
    """Voxelise int32 coords [N, 4] (vox_unique), the `strides` strided maps below it (map_stride, tensor strides 2, 4, ...),
    and for the first `tail_levels` levels the kernel_size-3 map onto itself plus the COUNT phase of its tail map -- all queued
    back to back on the current stream with every row count staying on the device (the *_dev entry points), followed by
    ONE host read of all sizes.  Same maps, bit for bit, as the call-by-call path (which reads each size as it is made:
    1 + strides + tail_levels reads); buffers are sized for the point count, which bounds every level.
    second_stream: the kernel maps and tail counts of the first levels run there, next to the strided maps of the deeper levels
    on the current stream (a level's kernel map needs only that level): the chain is all small latency-bound kernels, so the
    two halves overlap almost completely; the current stream joins before the read.
    on_level_dev(lv, rows_bound, d_count) (only with second_stream): called for every level under second_stream, BEHIND the event
    the read waits for -- work that needs a level's coordinates but not its size on the host (DiffCompletion: the part -> full
    matches, lidiff_nn_match_dev) runs there while the host is still blocked in the read.
    feed: a SizeFeed that receives this pyramid's sizes -- from the host read, or (read_free) from the device itself.
    read_free: NO host read.  Every level is handed over at its BOUND (the point count: coords [n, 4], tables of pitch n, tail
    maps pending with a pair bound) together with the device-side counts (Pyramid.counts); the sizes reach the host through
    `feed` (lidiff_publish_words) while later work is already queued.  The caller passes the counts on to every consumer
    (d_rows of spconv_fwd, ...), decides kernel choices from the sizes of an EARLIER pyramid and checks feed.validate() later."""
    require_device(coords, status)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    lib = _lib.load()
    _trace("enter")
    counts = torch.zeros(strides + 1 + tail_levels, dtype=torch.int32, device=dev)
    cptr = lambda i: counts.data_ptr() + 4 * i
    st = stream_ptr()
    # every table that starts as 0xFF bytes -- the hash tables of all levels, the kernel_size-3 tables of the tail levels -- in ONE
    # pool with ONE fill (preinit = 1 below); round 5: three clears per level and one per table
    pool = BytePool(dev)
    tables = [HashTable(n, dev, pool) for _ in range(strides + 1)]
    nbr_t = [pool.reserve((27, n), torch.int32) for _ in range(min(tail_levels, strides + 1))]
    pool.commit(0xFF)
    if second_stream is not None:
        pool.buf.record_stream(second_stream)
    nbr_t = [pool.view(t) for t in nbr_t]
    rows = [torch.empty((n, 4), dtype=torch.int32, device=dev)]
    first_idx = torch.empty(n, dtype=torch.int32, device=dev)
    inverse = torch.empty(n, dtype=torch.int64, device=dev)
    keep = [_workspace(n, dev)]
    call("lidiff_vox_unique", ptr(coords), n, ptr(tables[0].keys), ptr(tables[0].vals), tables[0].cap, ptr(rows[0]),
         ptr(first_idx), ptr(inverse), cptr(0), ptr(status), ptr(keep[0]), 1, st)
    _trace("level 0 queued")
    cur = torch.cuda.current_stream(dev)
    level_ev = []
    if second_stream is not None:
        level_ev.append(torch.cuda.Event())
        level_ev[0].record(cur)
    parents = [None]
    for lv in range(1, strides + 1):
        _trace(f"level {lv}: alloc")
        rows.append(torch.empty((n, 4), dtype=torch.int32, device=dev))
        parents.append(torch.empty(n, dtype=torch.int32, device=dev))
        keep.append(_workspace(n, dev))
        call("lidiff_map_stride_dev", ptr(rows[lv - 1]), n, cptr(lv - 1), 1 << lv, ptr(tables[lv].keys), ptr(tables[lv].vals),
             tables[lv].cap, ptr(rows[lv]), ptr(parents[lv]), cptr(lv), ptr(status), ptr(keep[lv]), 1, st)
        if second_stream is not None and lv < tail_levels:
            level_ev.append(torch.cuda.Event())
            level_ev[lv].record(cur)
    _trace("strides queued")
    if second_stream is not None and on_level_dev is not None:
        strides_done = torch.cuda.Event()
        strides_done.record(cur)
    tails = []
    tail_ctx = torch.cuda.stream(second_stream) if second_stream is not None else contextlib.nullcontext()
    with tail_ctx:
        st2 = stream_ptr()
        for lv in range(min(tail_levels, strides + 1)):
            if second_stream is not None:
                second_stream.wait_event(level_ev[lv])
            nbr = nbr_t[lv]
            call("lidiff_kernel_map_self_dev", ptr(rows[lv]), n, cptr(lv), ptr(tables[lv].keys), ptr(tables[lv].vals),
                 tables[lv].cap, 1 << lv, ptr(nbr), 1, st2)
            ws = torch.empty(lib.lidiff_tail_map_workspace_bytes(27, n), dtype=torch.uint8, device=dev)
            off = torch.empty(28, dtype=torch.int32, device=dev)
            row_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            call("lidiff_tail_map_dev", ptr(nbr), 27, n, cptr(lv), 13, ptr(off), ptr(row_ptr), 0, None, None, ptr(ws), st2)
            counts[strides + 1 + lv:strides + 2 + lv].copy_(off[27:28])
            tails.append((nbr, ws, off, row_ptr))
            if second_stream is not None:                   # allocated under second_stream, consumed on the current one
                for t_ in (nbr, ws, off, row_ptr):
                    t_.record_stream(cur)
        if second_stream is not None:
            joined = torch.cuda.Event()
            joined.record(second_stream)
            counts.record_stream(second_stream)         # (allocated on the current stream, read / written on the second)
            for lv in range(min(tail_levels, strides + 1)):
                for t in (rows[lv], tables[lv].keys, tables[lv].vals):
                    t.record_stream(second_stream)
            if on_level_dev is not None:
                second_stream.wait_event(strides_done)
                for lv in range(strides + 1):
                    rows[lv].record_stream(second_stream)
                    on_level_dev(lv, rows[lv], counts[lv:lv + 1])
    if second_stream is not None:
        cur.wait_event(joined)
    _trace("tails queued")
    if read_free:
        assert feed is not None
        feed.publish(counts, status)
        out = Pyramid()
        out.fast = out.feats0 = None
        out.counts, out.bound = counts, n
        out.coords = rows
        out.tables, out.inverse, out.first_idx = tables, inverse, first_idx
        out.parents = parents
        out.nbr3 = [t[0] for t in tails]
        out.tails = [TailMap(None, bounded=(nbr, n, counts[lv:lv + 1], ws, off, row_ptr, max(1, int(TAIL_PAIR_BOUND * n)), status))
                     for lv, (nbr, ws, off, row_ptr) in enumerate(tails)]
        _trace("done")
        return out
    # THE host read of this pyramid: into a pinned buffer, the host spinning on an event query (torch's tolist() goes through a
    # pageable staging copy and a blocking synchronise: ~0.1 ms later at the next launch; PINNED_READ = False keeps that form)
    if PINNED_READ:
        import threading
        key = (counts.numel(), str(dev), threading.get_ident())       # (one staging buffer per size, device AND thread: ADVICE r4)
        hb = _PINNED.get(key)
        if hb is None:
            hb = _PINNED[key] = torch.empty(counts.numel(), dtype=torch.int32, pin_memory=True)
        hb.copy_(counts, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        for _ in range(20000):                  # a short spin (the chain is ~0.4 ms away), then a blocking wait that frees the GIL
            if done.query():
                break
        else:
            done.synchronize()
        host = hb.tolist()
    else:
        host = counts.tolist()
    _trace("sizes read")
    if feed is not None:
        feed.push_host(host)
    out = Pyramid()
    out.fast = out.feats0 = None
    out.counts, out.bound = counts, n
    out.coords = [rows[lv][:host[lv]] for lv in range(strides + 1)]
    out.tables, out.inverse, out.first_idx = tables, inverse, first_idx[:host[0]]
    out.parents = [None] + [parents[lv][:host[lv - 1]] for lv in range(1, strides + 1)]
    out.nbr3, out.tails = [], []
    for lv, (nbr, ws, off, row_ptr) in enumerate(tails):
        m = host[lv]
        out.nbr3.append(nbr[:, :m].contiguous())               # the table at its own pitch (27 x M ints: a ~10 us copy)
        out.tails.append(TailMap(None, phase1=(nbr, n, counts[lv:lv + 1], ws, off, row_ptr, host[strides + 1 + lv], m)))
    _trace("done")
    return out


def build_pyramid_lanes(coords: torch.Tensor, status: torch.Tensor, feed: "SizeFeed", second_stream, third_stream,
                        strides: int = 4, on_level_dev=None, feats: torch.Tensor | None = None, tail_levels=(True, False),
                        up_pairs_levels=(), sorted_levels=()) -> Pyramid:
    """build_pyramid(read_free=True) with the work ordered by WHO WAITS FOR IT, and with EVERY map of LiDiff's networks queued
    up front (round 5).  Without a host read nothing about a map has to reach the host before it can be built, so nothing needs
    to be built "on demand" in the middle of the network any more (each such build stalled the network's queue for a chain of
    2-3 small kernels: 13 maps per step), and the chain behind x_t's points -- the critical path between two networks -- shrinks
    to what the stem needs.  Three lanes (tools/debug/event_gap.py: a small kernel cannot start while ANOTHER queue keeps the chip
    full, so what matters is which queue waits for what, not how much runs "in parallel"):
      current stream   voxel map of level 0 -> voxel mean of `feats` -> kernel_size-3 map of level 0 + its tail map (counted, and
                       filled when tail_levels[0]: a bounded fill needs no size on the host) -> event ready[1];
      third_stream     per level 1 .. strides: strided map -> kernel_size-2 map down into it -> its kernel_size-3 map (+ tail map
                       of level 1) -> event ready[2 ** level]; then the size record for the feed; then the decoder's maps, coarse
                       to fine: the transposed kernel_size-2 maps and (up_pairs_levels) their offset-grouped pair lists -> ready_up;
      second_stream    on_level_dev(level, rows, count) per level as soon as that level's rows exist (the part -> full matches).
    The current stream finally waits for the other lanes (anything built on it later is ordered behind the pyramid).
    Pyramid.ready = {tensor stride: event}, Pyramid.ready_up; Pyramid.down / up / up_lists hold the extra maps by level."""
    require_device(coords, status)
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4 and strides >= 1
    coords = coords.contiguous()
    n, dev = coords.shape[0], coords.device
    lib = _lib.load()
    cur = torch.cuda.current_stream(dev)
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    # Everything that starts as 0xFF bytes (hash tables, kernel_size-3 and kernel_size-2 neighbour tables of every level) lives in
    # ONE pool cleared by ONE fill, everything that starts as zeros (the counts, the voxel-mean workspace) in a second one: two
    # clears per pyramid where round 5 queued one per table (3 per hash table, 1 per neighbour table: ~30), VERDICT r5 #2
    pool, zpool = BytePool(dev), BytePool(dev)
    tables = [HashTable(n, dev, pool) for _ in range(strides + 1)]
    nbr3 = [pool.reserve((27, n), torch.int32) for _ in range(strides + 1)]
    down = [None] + [pool.reserve((8, n), torch.int32) for _ in range(strides)]           # down[lv]: level lv - 1 -> lv
    pool.commit(0xFF)
    nbr3, down = [pool.view(t) for t in nbr3], [None] + [pool.view(t) for t in down[1:]]
    counts = zpool.reserve((strides + 3,), torch.int32)
    mean_ws = zpool.reserve((lib.lidiff_vox_mean_workspace_bytes(n, feats.shape[1]),), torch.uint8) if feats is not None else None
    zpool.commit(0)
    counts, mean_ws = zpool.view(counts), (zpool.view(mean_ws) if mean_ws is not None else None)
    cptr = lambda i: counts.data_ptr() + 4 * i
    rows = [i32(n, 4) for _ in range(strides + 1)]
    parents = [None] + [i32(n) for _ in range(strides)]
    keep = [_workspace(n, dev) for _ in range(strides + 1)]
    first_idx, inverse = i32(n), torch.empty(n, dtype=torch.int64, device=dev)
    up = [None] + [i32(8, n) for _ in range(strides)]             # up[lv]:   level lv -> lv - 1 (output rows: level lv - 1)
    tail_buf = [(torch.empty(lib.lidiff_tail_map_workspace_bytes(27, n), dtype=torch.uint8, device=dev), i32(28), i32(n + 1))
                for _ in range(2)]
    shared = ([pool.buf, zpool.buf, first_idx, inverse] + rows + parents[1:] + keep + up[1:] + [q for tb in tail_buf for q in tb])
    for t in shared:           # allocated under the current stream; the other lanes read / write them
        t.record_stream(third_stream)
        t.record_stream(second_stream)

    def kmap3(lv):
        call("lidiff_kernel_map_self_dev", ptr(rows[lv]), n, cptr(lv), ptr(tables[lv].keys), ptr(tables[lv].vals), tables[lv].cap,
             1 << lv, ptr(nbr3[lv]), 1, stream_ptr())

    def tail(lv, fill):
        ws, off, row_ptr = tail_buf[lv]
        call("lidiff_tail_map_dev", ptr(nbr3[lv]), 27, n, cptr(lv), 13, ptr(off), ptr(row_ptr), 0, None, None, ptr(ws), stream_ptr())
        counts[strides + 1 + lv:strides + 2 + lv].copy_(off[27:28])
        tm = TailMap(None, bounded=(nbr3[lv], n, counts[lv:lv + 1], ws, off, row_ptr, max(1, int(TAIL_PAIR_BOUND * n)), status))
        return tm.fill() if fill else tm

    out = Pyramid()
    out.ready, out.up_lists = {}, {}
    # ---- lane 1 (current stream): what the stem needs
    call("lidiff_vox_unique", ptr(coords), n, ptr(tables[0].keys), ptr(tables[0].vals), tables[0].cap, ptr(rows[0]),
         ptr(first_idx), ptr(inverse), cptr(0), ptr(status), ptr(keep[0]), 1, stream_ptr())
    ev = [torch.cuda.Event()]
    ev[0].record(cur)
    out.feats0 = vox_mean(feats, inverse, n, zeroed_ws=mean_ws)[0] if feats is not None else None
    kmap3(0)
    tails = [tail(0, tail_levels[0])]
    out.ready[1] = out.fast = torch.cuda.Event()
    out.fast.record(cur)
    # ---- lane 3: level by level in the order the encoder walks them, the size record, then the decoder's maps
    third_stream.wait_event(ev[0])
    with torch.cuda.stream(third_stream):
        for lv in range(1, strides + 1):
            call("lidiff_map_stride_dev", ptr(rows[lv - 1]), n, cptr(lv - 1), 1 << lv, ptr(tables[lv].keys), ptr(tables[lv].vals),
                 tables[lv].cap, ptr(rows[lv]), ptr(parents[lv]), cptr(lv), ptr(status), ptr(keep[lv]), 1, stream_ptr())
            ev.append(torch.cuda.Event())
            ev[lv].record(third_stream)
            call("lidiff_kernel_map_down_dev", ptr(rows[lv - 1]), ptr(parents[lv]), n, cptr(lv - 1), 1 << (lv - 1), n, ptr(down[lv]),
                 1, stream_ptr())
            kmap3(lv)
            if lv == 1:
                tails.append(tail(1, tail_levels[1]))
            if lv in sorted_levels:            # the row order the split-operand kernel skips blocks under (kept on the table tensor)
                for t in mask_sorted_map(nbr3[lv]):
                    t.record_stream(cur)
            out.ready[1 << lv] = torch.cuda.Event()
            out.ready[1 << lv].record(third_stream)
        third_stream.wait_event(out.fast)                   # (level 0's tail count is part of the record)
        feed.publish(counts, status)
        for lv in range(strides, 0, -1):
            call("lidiff_kernel_map_up_dev", ptr(rows[lv - 1]), ptr(parents[lv]), n, cptr(lv - 1), 1 << (lv - 1), ptr(up[lv]), stream_ptr())
            if lv in up_pairs_levels:          # (pairs grouped by offset, the output-row order, the table's columns in that order)
                pin, order, off = rulebook_compact(up[lv], total=n, bounded=True)
                out.up_lists[lv] = (pin, order, off, up[lv].index_select(1, order.long()).contiguous())
        out.ready_up = torch.cuda.Event()
        out.ready_up.record(third_stream)
    # ---- lane 2: per-level callbacks (the matches), each behind its level only
    if on_level_dev is not None:
        with torch.cuda.stream(second_stream):
            for lv in range(strides + 1):
                second_stream.wait_event(ev[lv])
                on_level_dev(lv, rows[lv], counts[lv:lv + 1])
    cur.wait_event(out.ready_up)
    out.counts, out.bound = counts, n
    out.coords, out.tables, out.inverse, out.first_idx, out.parents = rows, tables, inverse, first_idx, parents
    out.nbr3, out.tails, out.down, out.up = nbr3, tails, down, up
    return out


def vox_mean(feats: torch.Tensor, inverse: torch.Tensor, m: int, zeroed_ws: torch.Tensor | None = None):
    """UNWEIGHTED_AVERAGE features (pipeline:77).  Returns (out[M,C], counts[M]).  Deterministic (fixed-point sums).
    zeroed_ws: a workspace the caller has already zeroed (part of a pyramid's zero pool: no clear of its own)."""
    require_device(feats, inverse)
    feats = feats.contiguous().float()
    n, c = feats.shape
    out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
    counts = torch.empty(m, dtype=torch.float32, device=feats.device)
    need = _lib.load().lidiff_vox_mean_workspace_bytes(int(m), int(c))
    ws = zeroed_ws if zeroed_ws is not None else torch.empty(need, dtype=torch.uint8, device=feats.device)
    assert ws.numel() >= need and ws.dtype == torch.uint8
    call("lidiff_vox_mean", ptr(feats), ptr(inverse), n, c, m, ptr(out), ptr(counts), ptr(ws), int(zeroed_ws is not None), stream_ptr())
    return out, counts


def vox_mean_bwd(grad_out, inverse, counts):
    grad_out = grad_out.contiguous()
    n, c = inverse.shape[0], grad_out.shape[1]
    g = torch.empty((n, c), dtype=torch.float32, device=grad_out.device)
    call("lidiff_vox_mean_bwd", ptr(grad_out), ptr(inverse), ptr(counts), n, c, ptr(g), stream_ptr())
    return g


def kernel_map(out_coords: torch.Tensor, in_table: HashTable, ks: int, step: int, self_map: bool = False) -> torch.Tensor:
    """Neighbour table nbr[K, M_out] (minkunet.py:53-66 ks=3; 13-29 ks=2/stride 2).  self_map: out_coords are the rows of
    in_table's own map and ks == 3 -- the mirrored offsets come for free (lidiff_kernel_map_self, same table)."""
    require_device(out_coords)
    out_coords = out_coords.contiguous()
    m = out_coords.shape[0]
    nbr = torch.empty((ks ** 3, m), dtype=torch.int32, device=out_coords.device)
    if self_map and ks == 3:
        call("lidiff_kernel_map_self", ptr(out_coords), m, ptr(in_table.keys), ptr(in_table.vals), in_table.cap, int(step),
             ptr(nbr), stream_ptr())
        return nbr
    call("lidiff_kernel_map", ptr(out_coords), m, ptr(in_table.keys), ptr(in_table.vals), in_table.cap,
         int(ks), int(step), ptr(nbr), stream_ptr())
    return nbr


def kernel_map_self_dev(coords_bound: torch.Tensor, d_rows: torch.Tensor, table: HashTable, step: int) -> torch.Tensor:
    """kernel_map(self_map=True) of a map whose row count lives on the device: coords_bound [n, 4] (the first d_rows[0] rows
    valid) -> nbr [27, n], columns behind the count -1."""
    require_device(coords_bound, d_rows)
    n = coords_bound.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=coords_bound.device)
    call("lidiff_kernel_map_self_dev", ptr(coords_bound), n, ptr(d_rows), ptr(table.keys), ptr(table.vals), table.cap, int(step),
         ptr(nbr), 0, stream_ptr())
    return nbr


def kernel_map_down_dev(fine_bound: torch.Tensor, parent_bound: torch.Tensor, d_rows_fine: torch.Tensor, ts_fine: int,
                        m_coarse_bound: int) -> torch.Tensor:
    """kernel_map_down with the fine map's row count on the device -> nbr [8, m_coarse_bound]."""
    require_device(fine_bound, parent_bound, d_rows_fine)
    assert parent_bound.dtype == torch.int32 and parent_bound.shape[0] == fine_bound.shape[0] and fine_bound.is_contiguous()
    nbr = torch.empty((8, m_coarse_bound), dtype=torch.int32, device=fine_bound.device)
    call("lidiff_kernel_map_down_dev", ptr(fine_bound), ptr(parent_bound), fine_bound.shape[0], ptr(d_rows_fine), int(ts_fine),
         int(m_coarse_bound), ptr(nbr), 0, stream_ptr())
    return nbr


def kernel_map_up_dev(fine_bound: torch.Tensor, parent_bound: torch.Tensor, d_rows_fine: torch.Tensor, ts_fine: int) -> torch.Tensor:
    """kernel_map_up with the row count on the device -> nbr [8, n], columns behind the count -1."""
    require_device(fine_bound, parent_bound, d_rows_fine)
    assert fine_bound.is_contiguous()
    n = fine_bound.shape[0]
    nbr = torch.empty((8, n), dtype=torch.int32, device=fine_bound.device)
    call("lidiff_kernel_map_up_dev", ptr(fine_bound), ptr(parent_bound), n, ptr(d_rows_fine), int(ts_fine), ptr(nbr), stream_ptr())
    return nbr


def kernel_map_down(fine_coords: torch.Tensor, parent: torch.Tensor, ts_fine: int, m_coarse: int) -> torch.Tensor:
    """The ks=2 / stride-2 table nbr[8, M_coarse] of a strided convolution (minkunet.py:13-29) from the parent array of
    map_stride -- the table kernel_map(coarse, fine table, 2, ts_fine) builds, without a lookup."""
    require_device(fine_coords, parent)
    fine_coords, parent = fine_coords.contiguous(), parent.contiguous()
    assert parent.dtype == torch.int32 and parent.shape[0] == fine_coords.shape[0]
    nbr = torch.empty((8, m_coarse), dtype=torch.int32, device=fine_coords.device)
    call("lidiff_kernel_map_down", ptr(fine_coords), ptr(parent), fine_coords.shape[0], int(ts_fine), int(m_coarse),
         ptr(nbr), stream_ptr())
    return nbr


def kernel_map_up(fine_coords: torch.Tensor, parent: torch.Tensor, ts_fine: int) -> torch.Tensor:
    """Neighbour table of the transposed ks=2/stride-2 conv (minkunet.py:32-46)."""
    require_device(fine_coords, parent)
    m = fine_coords.shape[0]
    nbr = torch.empty((8, m), dtype=torch.int32, device=fine_coords.device)
    call("lidiff_kernel_map_up", ptr(fine_coords.contiguous()), ptr(parent), m, int(ts_fine), ptr(nbr),
         stream_ptr())
    return nbr


def tile_order(coords: torch.Tensor, ts: int) -> torch.Tensor:
    """Permutation of a coordinate map's rows by Morton (Z-order) key at the map's resolution: consecutive rows
    of the result are neighbours in space, so a 128-row conv tile gathers overlapping input rows (L2 hits)."""
    require_device(coords)
    coords = coords.contiguous()
    m = coords.shape[0]
    keys = torch.empty(m, dtype=torch.int64, device=coords.device)
    call("lidiff_morton_keys", ptr(coords), m, int(ts), ptr(keys), stream_ptr())
    return torch.argsort(keys).to(torch.int32)


def rulebook_compact(nbr: torch.Tensor, total: int | None = None, bounded: bool = False):
    """ME-layout rulebook from a neighbour table: (pairs_in, pairs_out, offset_ptr[K+1]) -- pairs sorted by kernel offset,
    then by output row.  total: the number of pairs when the caller knows it (e.g. a transposed kernel_size-2 / stride-2 map
    has exactly one pair per output row) -- one pass, no host read; otherwise a counting pass, one read of the total, a fill pass.
    bounded (with total = the table's columns): a one-pair-per-row map whose valid columns are a device-side prefix of the table
    (the rest -1): the lists get `total` entries of which the first offset_ptr[K] are the pairs; pairs_out is completed to a
    PERMUTATION of the rows (the rows without a pair keep their own position), pairs_in behind the pairs is row 0."""
    require_device(nbr)
    k, m = nbr.shape
    dev = nbr.device
    ws = torch.empty(_lib.load().lidiff_rulebook_workspace_bytes(k, m), dtype=torch.uint8, device=dev)
    off = torch.empty(k + 1, dtype=torch.int32, device=dev)
    if total is None:
        call("lidiff_rulebook_compact", ptr(nbr), k, m, ptr(off), None, None, ptr(ws), stream_ptr())
        total = int(off[-1].item())
    if bounded:
        assert total == m
        pin = torch.zeros(total, dtype=torch.int32, device=dev)
        pout = torch.arange(total, dtype=torch.int32, device=dev)
    else:
        pin = torch.empty(total, dtype=torch.int32, device=dev)
        pout = torch.empty(total, dtype=torch.int32, device=dev)
    call("lidiff_rulebook_compact", ptr(nbr), k, m, ptr(off), ptr(pin), ptr(pout), ptr(ws), stream_ptr())
    return pin, pout, off


class ConvProfiler:
    """Optional per-launch HIP-event timing of the sparse-conv kernel (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream).  Only launches of
    the variants in `variants` are timed (None = all): every timed launch costs two event records in the
    timed region, so bench.py times the dominant variant only.  Pair counts are taken AFTER the run from the
    kept neighbour tables (no extra kernels in the timed region).
    Algorithmic work per launch: flops = 2*P*C_in*C_out,
    bytes = 4*(M_in*C_in + M_out*C_out) + 4*K*C_in*C_out + 8*P, P = valid pairs of the map (x replicas)."""

    def __init__(self, variants=None, sample: int = 1):
        """sample = n: only every n-th launch of a wanted variant carries events (n not a divisor of the launches per step, so
        that over the steps every layer is sampled): 1 / n of the event records in the timed region, the same average."""
        self.variants = None if variants is None else set(variants)
        self.sample = max(1, int(sample))
        self._seen = {}
        self.launches = []          # (variant, start|None, end|None, m_in, m_out, c_in, c_out, k, nbr|None, replicas)
        self.dw = []                # (start, end) of the weight-gradient launches (variant "dw"; bench.py's train leg)

    def dw_ms(self) -> float:
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.dw)

    def wants(self, variant: str) -> bool:
        if not (self.variants is None or variant in self.variants):
            return False
        if self.sample == 1 or variant == "dw":
            return True
        n = self._seen.get(variant, 0)
        self._seen[variant] = n + 1
        return n % self.sample == 0

    def summary(self):
        """{variant: dict(launches, ms, flops, bytes, timed, flops_timed, bytes_timed)} -- synchronises.  EVERY launch is
        counted in flops / bytes (shapes are host data, pair counts are taken after the run from the kept tables); `ms`,
        `flops_timed` and `bytes_timed` cover the launches that carried events (`timed` of them: all launches of the variants
        asked for, or every `sample`-th of them)."""
        torch.cuda.synchronize()
        counts = {}
        out = {}
        for variant, start, end, m_in, m_out, c_in, c_out, k, nbr, reps in self.launches:
            if nbr is None:
                p = m_out
            else:
                key = nbr.data_ptr()
                if key not in counts:
                    counts[key] = int((nbr >= 0).sum().item())
                p = counts[key]
            p, m_in, m_out = reps * p, reps * m_in, reps * m_out
            d = out.setdefault(variant, {"launches": 0, "timed": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0,
                                         "flops_timed": 0.0, "bytes_timed": 0.0, "mfma_flops_timed": 0.0})
            fl, by = 2.0 * p * c_in * c_out, 4.0 * (m_in * c_in + m_out * c_out) + 4.0 * k * c_in * c_out + 8.0 * p
            ex = fl
            if variant == "split3" and nbr is not None:
                # what the matrix pipe EXECUTES: six bf16 products per fp32 product, for every row of every 256-row tile under
                # every offset (the wide tiles multiply rows without a neighbour as zeros); rows = the centre offset's entries
                ck = ("rows", key)
                if ck not in counts:
                    counts[ck] = int((nbr[k // 2] >= 0).sum().item())
                ex = 6 * 2.0 * reps * (-(-counts[ck] // 256) * 256) * k * c_in * c_out
                # ... and what it executes WITH the block masks: the 16-row blocks that hold a neighbour under an offset
                bk = ("blocks", key)
                if bk not in counts:
                    mp = (nbr.shape[1] // 16) * 16
                    n_blk = int((nbr[:, :mp] >= 0).view(k, -1, 16).any(2).sum().item())
                    if mp < nbr.shape[1]:
                        n_blk += int((nbr[:, mp:] >= 0).any(1).sum().item())
                    counts[bk] = n_blk
                d["executed_flops_timed"] = d.get("executed_flops_timed", 0.0) + (6 * 2.0 * reps * 16 * counts[bk] * c_in * c_out if start is not None else 0.0)
            d["launches"] += 1
            if start is not None:
                d["timed"] += 1
                d["ms"] += start.elapsed_time(end)
                d["flops_timed"] += fl
                d["bytes_timed"] += by
                d["mfma_flops_timed"] += ex
            d["flops"] += fl
            d["bytes"] += by
        return out


def launches_by_bound(prof: "ConvProfiler", ridge: float, variants=None, tiny_rows: int = 32768):
    """The timed launches of `variants` (None = all) split by what bounds them: arithmetic intensity = algorithmic flops /
    algorithmic bytes of the launch against the machine's ridge point (peak FLOP/s / peak B/s) -- "mfma" at or
    above it, "hbm" below; launches with fewer than `tiny_rows` output rows (replicas included: at most one 128-row tile per
    compute unit -- the condition encoders' maps, the one-voxel unconditional branch) are "tiny": bound by launch latency, no
    roofline applies.  Per class: launches, ms, flops, bytes and the per-shape rows (variant, k, c_in, c_out) inside it."""
    torch.cuda.synchronize()
    counts = {}
    out = {b: {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "shapes": {}} for b in ("mfma", "hbm", "tiny")}
    for variant, start, end, m_in, m_out, c_in, c_out, k, nbr, reps in prof.launches:
        if start is None or (variants is not None and variant not in variants):
            continue
        if nbr is None:
            p = m_out
        else:
            key = nbr.data_ptr()
            if key not in counts:
                counts[key] = int((nbr >= 0).sum().item())
            p = counts[key]
        p, m_in, m_out = reps * p, reps * m_in, reps * m_out
        fl, by = 2.0 * p * c_in * c_out, 4.0 * (m_in * c_in + m_out * c_out) + 4.0 * k * c_in * c_out + 8.0 * p
        d = out["tiny" if m_out < tiny_rows else "mfma" if fl / by >= ridge else "hbm"]
        ms = start.elapsed_time(end)
        d["launches"] += 1
        d["ms"] += ms
        d["flops"] += fl
        d["bytes"] += by
        r = d["shapes"].setdefault((variant, k, c_in, c_out), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        r["launches"] += 1
        r["ms"] += ms
        r["flops"] += fl
        r["bytes"] += by
    return out


def layer_table(prof: "ConvProfiler", steps: int = 1):
    """Per-layer view of a profile in which every launch carried events: rows keyed by (variant, kernel volume, C_in, C_out,
    M_out bucket), with launches per step, ms per step, pairs per launch and algorithmic TFLOP/s -- where a step's time goes."""
    torch.cuda.synchronize()
    counts, rows = {}, {}
    for variant, start, end, m_in, m_out, c_in, c_out, k, nbr, reps in prof.launches:
        if start is None:
            continue
        if nbr is None:
            p = m_out
        else:
            key = nbr.data_ptr()
            if key not in counts:
                counts[key] = int((nbr >= 0).sum().item())
            p = counts[key]
        r = rows.setdefault((variant, k, c_in, c_out), {"launches": 0, "ms": 0.0, "flops": 0.0, "pairs": 0, "m_out": 0})
        r["launches"] += 1
        r["ms"] += start.elapsed_time(end)
        r["flops"] += 2.0 * reps * p * c_in * c_out
        r["pairs"] += reps * p
        r["m_out"] += reps * m_out
    out = []
    for (variant, k, c_in, c_out), r in rows.items():
        n = r["launches"]
        out.append({"variant": variant, "k": k, "c_in": c_in, "c_out": c_out, "launches_per_step": n / steps,
                    "ms_per_step": r["ms"] / steps, "avg_us": 1e3 * r["ms"] / n, "avg_pairs": r["pairs"] / n,
                    "avg_rows": r["m_out"] / n, "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0})
    return sorted(out, key=lambda d: -d["ms_per_step"])


PROFILER: ConvProfiler | None = None
# weight gradients: pair slices summed in slice order through a workspace (bit-reproducible); False = fp32 atomics
DETERMINISTIC_DW = True
# extra lidiff_spconv_fwd flag bits (include/lidiff_amd.h LIDIFF_CONV_*), e.g. 8 = LIDIFF_CONV_TILE_ONLY
CONV_FLAGS = 0


def conv_variant(c_out: int, kernel_id: int = 0) -> str:
    """Which kernel / template instantiation lidiff_spconv_fwd dispatches to (BN = output-channel tile of the tile kernel;
    "rows" = the streaming row kernel of identity maps, lidiff_spconv_fwd_kernel_id == 2)."""
    if kernel_id == 2:
        return "rows"
    if kernel_id == 3:
        return "thin"
    return ("bn128" if c_out % 128 == 0 else "bn96" if c_out % 96 == 0 else "bn64" if c_out % 64 == 0
            else "bn32" if c_out % 32 == 0 else "bn16")


def pack_weights(w: torch.Tensor) -> torch.Tensor:
    """[K, C_in, C_out] kernel -> MFMA fragment order (include/lidiff_amd.h, lidiff_spconv_pack_weights)."""
    require_device(w)
    w = w.detach().contiguous().float()
    k, c_in, c_out = w.shape
    n = _lib.load().lidiff_spconv_packed_weight_floats(k, c_in, c_out)
    wp = torch.empty(n, dtype=torch.float32, device=w.device)
    call("lidiff_spconv_pack_weights", ptr(w), k, c_in, c_out, ptr(wp), stream_ptr())
    return wp


def packed_weights(w: torch.Tensor, offset: int | None = None) -> torch.Tensor:
    """pack_weights(w) -- or, with `offset`, of the single kernel offset w[offset] as a K = 1 kernel -- cached on the
    tensor object (a module's Parameter) until it is modified in place, reallocated or moved: the pack costs one pass
    over the weights per optimizer step, not per forward."""
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    key = (w.data_ptr(), w._version, tuple(w3.shape), w.device)
    attr = "_lidiff_packed" if offset is None else f"_lidiff_packed_k{offset}"
    hit = getattr(w, attr, None)
    if hit is None or hit[0] != key:
        hit = (key, pack_weights(w3 if offset is None else w3[offset:offset + 1]))
        try:
            setattr(w, attr, hit)
        except AttributeError:      # not attachable: pack every call
            pass
    return hit[1]


def invalidate_caches(module: torch.nn.Module) -> None:
    """Drop everything derived from a module's tensors: the packed conv weights (keyed on the Parameter's
    data_ptr / _version) and the folded eval-BatchNorm scale / shift of the fused plan (keyed on _version).  Both
    keys follow ordinary in-place updates (optimizer steps, load_state_dict, copy_); writes THROUGH ``.data``
    (``p.data.mul_(...)``: an EMA, weight clipping) do not bump ``_version`` -- call this after such writes."""
    for p in module.parameters():
        for attr in [a for a in vars(p) if a.startswith("_lidiff_packed")]:
            delattr(p, attr)
    for m in module.modules():
        if hasattr(m, "_affine_cache"):
            del m._affine_cache


def spconv_fwd(in_a: torch.Tensor, w: torch.Tensor, nbr: torch.Tensor | None, m_out: int,
               in_b: torch.Tensor | None = None, scale=None, shift=None, residual=None,
               relu: bool = False, sparse_map: bool = False, replicas: int = 1,
               row_order: torch.Tensor | None = None, kernel: str | None = None, tail=None,
               offset: int | None = None, d_rows: torch.Tensor | None = None, rows_hint: int | None = None,
               in_rows_hint: int | None = None) -> torch.Tensor:
    """Sparse convolution forward with fused epilogue (MinkowskiConvolution[Transpose];
    minkunet.py:17,36,53,61,72).  w: [K, C_in, C_out] ([C_in, C_out] accepted for K == 1).  sparse_map: hint that
    the kernel map has only a few pairs per offset and 128-row tile (CoordinateManager.is_sparse_map).
    replicas: R feature matrices stacked row-wise share the map and the weights (the CFG pair): in_a is
    [R * M_in, C], the result [R * m_out, C_out].
    row_order: int32 permutation of the output rows (tile_order()); `nbr` must then hold its columns in that
    order (nbr[:, row_order]).  Results do not depend on it.
    kernel: "tile_only" keeps identity maps off the row kernel (spconv_rows.hip); "tile128" keeps 64-column layers on 128-row
    tiles (large maps take 256-row tiles by default).
    d_rows: int32 [1] on the device -- the number of VALID output rows when m_out is only their bound (a step without host reads):
    m_out still shapes the result, the table's pitch and the replica pitch; tiles behind the count leave at once.  rows_hint: the
    row count the host BELIEVES (an earlier step's): it alone picks the tile size where the bound would pick another one.
    offset: convolve with the single kernel offset w[offset] over the identity map (nbr must be None): the centre of a
    kernel_size-3 map.  tail = (rows [R * P, C_out], ptr int32 [m_out + 1], idx int32 [P]): rows added to the sum before
    the epilogue, out[o] += sum(rows[idx[ptr[o]:ptr[o + 1]]]) -- the other offsets' contributions (TailMap)."""
    require_device(in_a, w, nbr, in_b, scale, shift, residual)
    wp = packed_weights(w, offset)
    if w.dim() == 2:
        k, (c_in, c_out) = 1, w.shape
    else:
        k, c_in, c_out = w.shape
    if offset is not None:
        assert nbr is None and 0 <= offset < k
        k = 1
    in_a = in_a.contiguous()
    c_a = in_a.shape[1]
    c_b = 0
    if in_b is not None:
        in_b = in_b.contiguous()
        c_b = in_b.shape[1]
        assert in_b.shape[0] == in_a.shape[0]
    assert c_a + c_b == c_in, f"channel mismatch {c_a}+{c_b} != {c_in}"
    assert replicas >= 1 and in_a.shape[0] % replicas == 0
    m_in = in_a.shape[0] // replicas
    if nbr is not None:
        assert nbr.shape == (k, m_out) and nbr.dtype == torch.int32 and nbr.is_contiguous()
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (replicas * m_out, c_out)
    if row_order is not None:
        assert row_order.dtype == torch.int32 and row_order.shape == (m_out,) and row_order.is_contiguous()
    t_rows = t_ptr = t_idx = None
    n_tail = 0
    if tail is not None:
        t_rows, t_ptr, t_idx = tail
        n_tail = t_idx.shape[0]
        assert t_rows.shape == (replicas * n_tail, c_out) and t_rows.is_contiguous() and t_rows.dtype == torch.float32
        assert t_ptr.dtype == torch.int32 and t_ptr.shape == (m_out + 1,) and t_idx.dtype == torch.int32
    out = torch.empty((replicas * m_out, c_out), dtype=torch.float32, device=in_a.device)
    flags = int(bool(sparse_map)) | {"tile": 0, "tile_only": 8, "tile128": 16}[kernel or "tile"] | CONV_FLAGS
    if rows_hint is not None and rows_hint * replicas < 256 * 512:
        flags |= 16                 # the exact-size path would keep 128-row tiles for this map: the same choice under a bound
    prof = PROFILER
    variant = None
    if prof is not None:
        variant = conv_variant(c_out, _lib.load().lidiff_spconv_fwd_kernel_id(c_a, c_b, c_out, k, int(nbr is not None),
                                                                             int(row_order is not None), flags))
    timed = prof is not None and prof.wants(variant)
    start = end = None
    if timed:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.join_pending()             # (the wait for maps still being built on a side stream is not the kernel's time)
        start.record()
    call("lidiff_spconv_fwd", ptr(in_a), c_a, ptr(in_b), c_b, ptr(wp), ptr(nbr), k, m_in, m_out,
         c_out, ptr(out), ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), ptr(row_order), int(replicas), flags,
         ptr(t_rows), ptr(t_ptr), ptr(t_idx), n_tail, ptr(d_rows), stream_ptr())
    if timed:
        end.record()
    if prof is not None:
        # (algorithmic bytes of a launch under a bound: from the rows the host believes, not from the bound)
        prof.launches.append((variant, start, end, m_in if in_rows_hint is None else int(in_rows_hint),
                              m_out if rows_hint is None else int(rows_hint), c_in, c_out, k, nbr, replicas))
    return out


# training-mode BatchNorm through the HIP kernels of norm.hip (False: torch's batch_norm kernels)
FUSED_BN_TRAIN = True


def bn_train_applies(x: torch.Tensor) -> bool:
    return (FUSED_BN_TRAIN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= 2
            and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 1024)


def bn_sync_applies(x: torch.Tensor) -> bool:
    """The synchronised form of bn_train_applies(): decided from properties that are THE SAME ON EVERY RANK (dtype, width) and
    never from the local row count -- a rank holding 0 or 1 rows of some layer must issue the same collective as the others
    (a rank-local choice between this path's all-reduce and torch's all-gather-based function deadlocks the group);
    the kernels take any row count, the statistics are the group's."""
    return (FUSED_BN_TRAIN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
            and x.shape[1] % 4 == 0 and 4 <= x.shape[1] <= 1024)


def bn_fused_applies(bn, x: torch.Tensor) -> bool:
    """Does batch_norm_train(x, bn) apply: the rank-invariant rule for a SyncBatchNorm1d that shares statistics, else the local one."""
    if isinstance(bn, SyncBatchNorm1d) and bn.training and bn.group() is not None:
        return bn_sync_applies(x)
    return bn_train_applies(x)


class SyncBatchNorm1d(torch.nn.BatchNorm1d):
    """The `.bn` child of ME.MinkowskiSyncBatchNorm (train.py:90 ``convert_sync_batchnorm``): an nn.BatchNorm1d -- same
    parameters, buffers and state-dict keys -- whose TRAINING statistics are taken over every rank of `process_group`.  The
    normalisation itself stays on the kernels of norm.hip (_BatchNormTrain with group=...): per-channel (sum x, sum x^2, rows)
    leave the device reduction as fp64, ONE all-reduce of 2C + 1 doubles per layer shares them (RCCL under "nccl"; an fp64 SUM
    gives every rank the same bits), backward shares (sum dy, sum dy (x - mean)) the same way.  Eval mode, CPU tensors and
    widths the kernels do not take fall back to the local nn.BatchNorm1d arithmetic / torch's SyncBatchNorm function."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine, track_running_stats=track_running_stats)
        self.process_group = process_group

    # True: take the synchronised path (sums -> all-reduce -> statistics) even in a group of ONE rank -- what bench.py's train leg
    # times on the 1-GPU box: every launch and collective of the multi-rank step, minus the wire
    sync_single_rank = False

    def group(self):
        """The process group to share statistics over, or None when there is nobody to share with."""
        import torch.distributed as tdist
        if not (tdist.is_available() and tdist.is_initialized()):
            return None
        g = self.process_group if self.process_group is not None else tdist.group.WORLD
        return g if (tdist.get_world_size(g) > 1 or self.sync_single_rank) else None

    def forward(self, x):
        g = self.group() if self.training else None
        if g is None:
            return super().forward(x)
        if bn_sync_applies(x) and self.momentum is not None and torch.is_grad_enabled():
            return batch_norm_train(x, self)
        # shapes norm.hip does not take: torch's own synchronised function (same statistics, its kernels)
        from torch.nn.modules._functions import SyncBatchNorm as _TorchSync
        import torch.distributed as tdist
        if self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
        return _TorchSync.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                0.1 if self.momentum is None else self.momentum, g, tdist.get_world_size(g))


def bn_module_fused(bn) -> bool:
    """Is `bn` a BatchNorm1d whose training pass batch_norm_train() implements (plain, or the synchronised subclass)?"""
    return (type(bn) in (torch.nn.BatchNorm1d, SyncBatchNorm1d) and bn.training and bn.momentum is not None
            and torch.is_grad_enabled())


class _BatchNormTrain(torch.autograd.Function):
    """Training-mode nn.BatchNorm1d on a feature matrix [M, C] (MinkowskiBatchNorm, minkunet.py:23) through lidiff_bn_stats /
    lidiff_bn_apply / lidiff_bn_bwd: batch statistics with double accumulators in a fixed order (deterministic), running
    estimates updated as torch does (momentum, unbiased variance).  relu: the MinkowskiReLU that follows, fused; residual: the
    ResidualBlock's shortcut added in front of that ReLU (minkunet.py:79: relu(net(x) + downsample(x))), fused as well.
    group: a torch.distributed process group -- SyncBatchNorm (train.py:90): the statistics of forward and backward are summed
    over its ranks (lidiff_bn_sums / _stats_from_sums / _bwd_sums / _bwd_apply around one all-reduce each)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu, residual=None, group=None):
        x = x.contiguous()
        residual = None if residual is None else residual.contiguous()
        m, c = x.shape
        dev = x.device
        stats = torch.empty((3, c), dtype=torch.float32, device=dev)
        ws = torch.empty(_lib.load().lidiff_bn_workspace_bytes(c), dtype=torch.uint8, device=dev)
        count = None
        rm = running_mean if running_mean is not None and running_mean.dtype == torch.float32 else None
        rv = running_var if rm is not None else None
        if group is None:
            call("lidiff_bn_stats", ptr(x), m, c, float(eps), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(rm), ptr(rv),
                 float(momentum), ptr(ws), stream_ptr())
        else:
            import torch.distributed as tdist
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            call("lidiff_bn_sums", ptr(x), m, c, ptr(sums), ptr(ws), stream_ptr())         # sums[2c] = m
            tdist.all_reduce(sums, op=tdist.ReduceOp.SUM, group=group)
            call("lidiff_bn_stats_from_sums", ptr(sums), c, float(eps), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(rm), ptr(rv),
                 float(momentum), stream_ptr())
            count = sums[2 * c:]                                 # [1] fp64, stays on the device
        if rm is not None:       # updated inside the statistics launch, behind autograd's back: bump the version counters by hand
            torch.autograd.graph.increment_version(running_mean)      # (they key the folded eval-mode scale / shift caches)
            torch.autograd.graph.increment_version(running_var)
        y = torch.empty_like(x)
        w = None if weight is None else weight.detach().float().contiguous()
        b = None if bias is None else bias.detach().float().contiguous()
        # bf16 training: a normalise + ReLU output goes into a convolution next (the block's second convolution, the next block's
        # first) -- its bf16 shadow is written by the same launch.  Without the ReLU the output is a shortcut on its way into an
        # add: no shadow (a convolution that does gather such a tensor casts it itself, ops.cast_bf16)
        ctx.shadow = TRAIN_OPERANDS == "bf16" and BF16_ROWS and c % 32 == 0
        y16 = torch.empty(x.shape, dtype=torch.bfloat16, device=dev) if ctx.shadow and relu else None
        call("lidiff_bn_apply", ptr(x), m, c, ptr(stats[0]), ptr(stats[2]), ptr(w), ptr(b), ptr(residual), int(bool(relu)), ptr(y),
             ptr(y16), stream_ptr())
        if y16 is not None:
            y._lidiff_bf16 = ((y.data_ptr(), y._version, tuple(y.shape)), y16)
        if running_mean is not None and rm is None:              # running estimates of another dtype: torch's own arithmetic
            with torch.no_grad():
                cnt = float(m) if count is None else count
                running_mean.mul_(1.0 - momentum).add_(stats[0], alpha=momentum)
                running_var.mul_(1.0 - momentum).add_(stats[1] * (cnt / (cnt - 1.0)) * momentum)
        ctx.save_for_backward(x, w, stats, y if relu else None, count)
        ctx.relu = bool(relu)
        ctx.has_affine = (weight is not None, bias is not None)
        ctx.has_residual = residual is not None
        ctx.group = group
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stats, y, count = ctx.saved_tensors
        dy = dy.contiguous()
        m, c = x.shape
        sums = torch.empty((2, c), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dx16 = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if dx is not None and ctx.shadow else None
        dres = None
        if ctx.has_residual and ctx.needs_input_grad[8]:
            dres = torch.empty_like(x) if ctx.relu else dy      # without ReLU the residual's gradient is dy itself
        ws = torch.empty(_lib.load().lidiff_bn_workspace_bytes(c), dtype=torch.uint8, device=x.device)
        if ctx.group is None:
            call("lidiff_bn_bwd", ptr(dy), ptr(x), ptr(y), m, c, ptr(stats[0]), ptr(stats[2]), ptr(w), ptr(sums[0]), ptr(sums[1]),
                 ptr(dx), ptr(dres) if ctx.relu else None, ptr(ws), ptr(dx16), stream_ptr())
            local = sums
        else:
            import torch.distributed as tdist
            dsums = torch.empty(2 * c, dtype=torch.float64, device=x.device)
            call("lidiff_bn_bwd_sums", ptr(dy), ptr(x), ptr(y), m, c, ptr(stats[0]), ptr(dsums), ptr(ws), stream_ptr())
            local = dsums.float().view(2, c)                     # THIS rank's sums: d beta / d gamma (the gradient all-reduce averages them)
            tdist.all_reduce(dsums, op=tdist.ReduceOp.SUM, group=ctx.group)
            call("lidiff_bn_bwd_apply", ptr(dy), ptr(x), ptr(y), m, c, ptr(stats[0]), ptr(stats[2]), ptr(w), ptr(dsums), ptr(count),
                 ptr(sums[0]), ptr(sums[1]), ptr(dx), ptr(dres) if ctx.relu else None, ptr(dx16), stream_ptr())
        if dx16 is not None:         # (the gradient of the convolution in front of this layer: it gathers the shadow)
            dx._lidiff_bf16 = ((dx.data_ptr(), dx._version, tuple(dx.shape)), dx16)
        dw = local[1] * stats[2] if ctx.has_affine[0] and ctx.needs_input_grad[1] else None
        db = local[0].clone() if ctx.has_affine[1] and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None, None, None, dres, None


def batch_norm_train(x, bn: torch.nn.BatchNorm1d, relu: bool = False, residual=None):
    """bn(x) in training mode (optionally followed by ReLU) through _BatchNormTrain; counts the batch like torch does.
    A SyncBatchNorm1d shares its statistics over its process group."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    rm, rv = (bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None)
    group = bn.group() if isinstance(bn, SyncBatchNorm1d) else None
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, rm, rv, momentum, bn.eps, relu, residual, group)


def pairs_kernel_applies(c_a: int, c_b: int, c_out: int) -> bool:
    """Shapes lidiff_spconv_fwd_pairs takes (input widths multiples of 16 summing to 32 / 64 / 96 / 128, C_out % 32 == 0)."""
    return bool(_lib.load().lidiff_spconv_fwd_pairs_supported(c_a, c_b, c_out)) and not (CONV_FLAGS & 8)


def spconv_fwd_pairs(in_a: torch.Tensor, w: torch.Tensor, pair_in: torch.Tensor, pair_out: torch.Tensor | None,
                     offset_ptr: torch.Tensor, m_out: int, in_b: torch.Tensor | None = None, scale=None, shift=None,
                     residual=None, relu: bool = False, replicas: int = 1, rows_hint: int | None = None,
                     in_rows_hint: int | None = None) -> torch.Tensor:
    """Sparse convolution over a map in which every output row has exactly ONE pair, given as a pair list grouped by kernel
    offset (lidiff_spconv_fwd_pairs; include/lidiff_amd.h): out[pair_out[p]] = epilogue(in[pair_in[p]] @ w[offset of p]),
    offset_ptr [K + 1] the pair ranges of the offsets; pair_out None = one output row per pair, in list order (then m_out ==
    number of pairs).  The transposed kernel_size-2 / stride-2 convolutions (MinkowskiConvolutionTranspose, minkunet.py:36) and
    the tail pass of spconv_centre_tail.  Same epilogue / replica arguments as spconv_fwd; bit-identical results."""
    require_device(in_a, w, pair_in, pair_out, offset_ptr, in_b, scale, shift, residual)
    wp = packed_weights(w)
    k, c_in, c_out = w.shape
    in_a = in_a.contiguous()
    c_a, c_b = in_a.shape[1], 0
    if in_b is not None:
        in_b = in_b.contiguous()
        c_b = in_b.shape[1]
    assert c_a + c_b == c_in and in_a.shape[0] % replicas == 0
    m_in = in_a.shape[0] // replicas
    n = pair_in.shape[0]
    assert pair_in.dtype == torch.int32 and offset_ptr.dtype == torch.int32 and offset_ptr.shape == (k + 1,)
    assert pair_out is None or (pair_out.dtype == torch.int32 and pair_out.shape == (n,))
    assert pair_out is not None or m_out == n
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (replicas * m_out, c_out)
    out = torch.empty((replicas * m_out, c_out), dtype=torch.float32, device=in_a.device)
    prof = PROFILER
    timed = prof is not None and prof.wants("rows")
    start = end = None
    if timed:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.join_pending()             # (the wait for maps still being built on a side stream is not the kernel's time)
        start.record()
    call("lidiff_spconv_fwd_pairs", ptr(in_a), c_a, ptr(in_b), c_b, ptr(wp), k, ptr(pair_in), ptr(pair_out), ptr(offset_ptr),
         n, m_in, m_out, c_out, ptr(out), ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), int(replicas), stream_ptr())
    if timed:
        end.record()
    if prof is not None:
        prof.launches.append(("rows", start, end, m_in if in_rows_hint is None else int(in_rows_hint),
                              m_out if rows_hint is None else int(rows_hint), c_in, c_out, k, None, replicas))
    return out


# GEMM operand precision of the TRAINING convolutions (_SparseConv forward and input gradient): "f32" or "bf16"
# (BASELINE.json configs[4]: train.py under bf16 autocast).  The inference path always runs fp32 (configs[1]).
TRAIN_OPERANDS = "f32"


class train_operands:
    """with ops.train_operands("bf16"): ... -- the precision of the training convolutions' GEMM operands."""

    def __init__(self, kind: str):
        assert kind in ("f32", "bf16")
        self.kind = kind

    def __enter__(self):
        global TRAIN_OPERANDS
        self.prev, TRAIN_OPERANDS = TRAIN_OPERANDS, self.kind

    def __exit__(self, *exc):
        global TRAIN_OPERANDS
        TRAIN_OPERANDS = self.prev


# The bf16 kernel has no packed stages.  Rounds 2-4: on low-density maps (the managers' sparse-map hint) the fp32 tile kernel was
# the faster one (profiles/r02_bf16_conv_sweep.txt) and those layers kept it.  Round 5: with bf16 shadow rows, W fragments through
# LDS and the 16-byte flush the bf16 kernel wins there too in aggregate (training step 120.3 -> 117.9 ms), so EVERY eligible layer
# runs in bf16 -- which is also what the oracle's emulation of the bf16 step assumes ("0": the former rule).
BF16_SPARSE_MAPS = True


def bf16_conv_applies(c_a: int, c_b: int, c_out: int, sparse_map: bool = False) -> bool:
    """Shapes lidiff_spconv_fwd_bf16 takes (every MinkUNet layer but the 3-channel stem and the 96 -> 3 head)."""
    return c_a % 32 == 0 and c_b % 32 == 0 and c_out % 32 == 0 and (BF16_SPARSE_MAPS or not sparse_map)


def packed_weights_bf16(w: torch.Tensor, transposed: bool = False, flip: bool = False, planes: int = 1) -> torch.Tensor:
    """lidiff_spconv_pack_weights_bf16 of w [K, C_in, C_out] -- or, transposed, of the input-gradient kernel
    w[::-1 if flip].transpose(1, 2) -- in `planes` bf16 pieces, cached on the Parameter like packed_weights()."""
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    key = (w.data_ptr(), w._version, tuple(w3.shape), w.device)
    attr = f"_lidiff_packed_bf16x{planes}" + ("_t" if transposed else "") + ("f" if flip else "")
    hit = getattr(w, attr, None)
    if hit is None or hit[0] != key:
        src = w3.detach()
        if transposed:
            src = (src.flip(0) if flip else src).transpose(1, 2)
        src = src.contiguous().float()
        k, c_in, c_out = src.shape
        n = _lib.load().lidiff_spconv_packed_weight_bf16_elems(k, c_in, c_out, planes)
        wp = torch.empty(n, dtype=torch.bfloat16, device=w.device)
        call("lidiff_spconv_pack_weights_bf16", ptr(src), k, c_in, c_out, planes, ptr(wp), stream_ptr())
        hit = (key, wp)
        try:
            setattr(w, attr, hit)
        except AttributeError:
            pass
    return hit[1]


# bf16 training: the convolutions gather bf16 SHADOW rows of their inputs (cast once per tensor, used by the forward, the input
# gradient and the weight gradient) instead of rounding the fp32 rows inside every kernel (False: as rounds 2-4)
BF16_ROWS = True


BF16_ROWS_KERNEL = {None: 1, "ring": 2, "two_stage": 3, "wide": 4}
# bf16 rows: kernel_size-3 and kernel_size-1 layers on the wide register-tile kernel (256-row tiles, no pair lists, one fp32 sum
# per output over all offsets and channels: 256 -> 256 at stride 8 738 -> 378 us, 96 -> 96 at stride 2 201 -> 102, 32 -> 64 120 ->
# 45); the stride-2 maps (8 offsets, 1-2.6 pairs per row: 22 vs 28 us, 55 vs 64) keep the pair-list kernels.  False: the two-stage /
# ring kernels everywhere, whose sums -- per offset first -- are bit-identical to the fp32-row form
BF16_WIDE = True


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """bf16 shadow of a contiguous fp32 matrix (round to nearest even, lidiff_cast_bf16), kept on the tensor object: the layers
    that read the same tensor (a block's first convolution and its shortcut; a convolution's forward and its weight gradient)
    share one copy."""
    require_device(x)
    hit = getattr(x, "_lidiff_bf16", None)
    if hit is not None and hit[0] == (x.data_ptr(), x._version, tuple(x.shape)):
        return hit[1]
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("lidiff_cast_bf16", ptr(x), x.numel(), ptr(out), stream_ptr())
    try:
        x._lidiff_bf16 = ((x.data_ptr(), x._version, tuple(x.shape)), out)
    except AttributeError:
        pass
    return out


def spconv_fwd_bf16(in_a: torch.Tensor, w: torch.Tensor, nbr: torch.Tensor | None, m_out: int,
                    in_b: torch.Tensor | None = None, scale=None, shift=None, residual=None, relu: bool = False,
                    replicas: int = 1, transposed: bool = False, flip: bool = False, planes: int = 1,
                    kernel: str | None = None) -> torch.Tensor:
    """spconv_fwd with bf16 matrix operands and fp32 accumulation (lidiff_spconv_fwd_bf16; include/lidiff_amd.h).
    kernel (bf16 rows only): "ring" / "two_stage" force one of the two bit-identical kernels (default: the library's choice).
    planes = 1: operands rounded to bf16 (mixed-precision training); planes = 2 / 3: every operand cut into 2 / 3 bf16
    pieces, 3 / 6 MFMAs per block, fp32-accurate results.
    transposed: convolve with w[::-1 if flip].transpose(1, 2) -- the input gradient over the swapped map."""
    require_device(in_a, w, nbr, in_b, scale, shift, residual)
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    k, c_in, c_out = w3.shape
    if transposed:
        c_in, c_out = c_out, c_in
    wp = packed_weights_bf16(w, transposed, flip, planes)
    in_a = in_a.contiguous()
    c_a, c_b = in_a.shape[1], 0
    if in_b is not None:
        in_b = in_b.contiguous()
        c_b = in_b.shape[1]
    assert c_a + c_b == c_in, f"channel mismatch {c_a}+{c_b} != {c_in}"
    rows16 = in_a.dtype == torch.bfloat16                # bf16 shadow rows (cast_bf16): planes = 1 only
    assert (in_a.dtype == torch.float32 or (rows16 and planes == 1)) and in_a.shape[0] % replicas == 0
    assert in_b is None or in_b.dtype == in_a.dtype
    m_in = in_a.shape[0] // replicas
    if nbr is not None:
        assert nbr.shape == (k, m_out) and nbr.dtype == torch.int32 and nbr.is_contiguous()
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (replicas * m_out, c_out)
    out = torch.empty((replicas * m_out, c_out), dtype=torch.float32, device=in_a.device)
    prof = PROFILER
    variant = "bf16" if planes == 1 else f"bf16x{planes}"
    timed = prof is not None and prof.wants(variant)
    start = end = None
    if timed:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.join_pending()             # (the wait for maps still being built on a side stream is not the kernel's time)
        start.record()
    call("lidiff_spconv_fwd_bf16", ptr(in_a), c_a, ptr(in_b), c_b, ptr(wp), int(planes), ptr(nbr), k, m_in, m_out, c_out,
         ptr(out), ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), int(replicas),
         ((4 if kernel is None and BF16_WIDE and k != 8 else BF16_ROWS_KERNEL[kernel]) if rows16 else 0), stream_ptr())
    if timed:
        end.record()
    if prof is not None:
        prof.launches.append((variant, start, end, m_in, m_out, c_in, c_out, k, nbr, replicas))
    return out


# Dense kernel_size-3 convolutions of the eval-mode fused plan (tensor stride >= SPLIT3_MIN_STRIDE, maps that are not low-density,
# widths lidiff_spconv_fwd_split3 takes) through the three-way split bf16 kernel: fp32 in, fp32 out, fp32 accuracy (include/lidiff_amd.h).
# LIDIFF_SPLIT3=0: every layer on the native fp32-MFMA kernel.
SPLIT3 = os.environ.get("LIDIFF_SPLIT3", "1") != "0"
SPLIT3_MIN_STRIDE = int(os.environ.get("LIDIFF_SPLIT3_MIN_STRIDE", "4"))
SPLIT3_K1_MIN_CIN = 256         # kernel_size-1 shortcuts narrower than this stay on the streaming row kernel (too little work per tile)
SPLIT3_MIN_TILES = 256          # fewer 256 x 128 tiles than compute units (the condition encoders, late steps' coarse levels): native kernel
# ... with the rows of the map sorted by their neighbour sets (mask_sorted_map), so that the kernel skips whole 16-row blocks
SPLIT3_SORTED = os.environ.get("LIDIFF_SPLIT3_SORTED", "1") != "0"
# Where a table is sorted: at its first use, on the consumer's stream (3 x ~0.1 ms per step on the denoiser's stream).  Sorting on
# the map lane of a pyramid built in lanes (True) takes it off that stream, but made the loop with side streams differ from the
# serial loop in the last bit of a few points once in ~10 runs (tests/test_gpu_network.py::test_overlapped_coordinate_pipeline_
# equals_the_serial_one; neither an event of the sort itself nor record_stream on its results cured it, a device synchronise did):
# off until that is understood.
SPLIT3_PRESORT = False


def split3_layer(tensor_stride: int, rows: int, replicas: int, c_in_a: int, c_in_b: int, c_out: int, m_bound: int | None = None) -> bool:
    """Does the fused plan run a kernel_size-3 convolution of this shape on the split-operand kernel?  (rows: what the host
    believes -- exact, or the same level of the role's previous pyramid; m_bound: the rows a replica's matrices are allocated for --
    the split matrices are addressed through 2 GiB buffer descriptors, a larger one keeps the native kernel)"""
    if m_bound is not None and m_bound * max(c_in_a, c_in_b) * 2 * SPLIT_PIECES >= (1 << 31):
        return False
    return (SPLIT3 and tensor_stride >= SPLIT3_MIN_STRIDE and split3_conv_applies(c_in_a, c_in_b, c_out)
            and -(-rows * replicas // 256) * (c_out // (128 if c_out % 128 == 0 else 64)) >= SPLIT3_MIN_TILES)


# Pieces per operand of the split-operand kernel: 3 (bf16; fp32 accuracy: the default) or 2 (fp16; 22-bit operands, three products
# instead of six -- opt-in: a reduction of precision, DESIGN.md 4.3, profiles/r06_f16x2.txt)
SPLIT_PIECES = int(os.environ.get("LIDIFF_SPLIT_PIECES", "3"))


class split_pieces:
    """Context manager: `with ops.split_pieces(2):` runs the split-operand layers on two fp16 pieces (A/B measurements, tests)."""

    def __init__(self, n: int):
        assert n in (2, 3)
        self.n = int(n)

    def __enter__(self):
        global SPLIT_PIECES
        self.prev, SPLIT_PIECES = SPLIT_PIECES, self.n

    def __exit__(self, *exc):
        global SPLIT_PIECES
        SPLIT_PIECES = self.prev


class split3:
    """Context manager: `with ops.split3(False):` runs the native fp32 kernel everywhere (A/B measurements, parity tests)."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global SPLIT3
        self.prev, SPLIT3 = SPLIT3, self.on

    def __exit__(self, *exc):
        global SPLIT3
        SPLIT3 = self.prev


def split3_rows(x: torch.Tensor, pieces: int | None = None, d_rows: torch.Tensor | None = None, replicas: int = 1) -> torch.Tensor:
    """fp32 rows [M, C] -> bf16 [M, 3, C] (pieces = 3: every value as the exact sum of three bf16 pieces) or fp16 [M, 2, C] (pieces = 2:
    22 bits of every value) (lidiff_split3_rows) -- the operand layout of spconv_fwd_split3.  Kept on the tensor object
    (`_lidiff_split3`): consumers of one tensor share the cut."""
    pieces = SPLIT_PIECES if pieces is None else int(pieces)
    hit = getattr(x, "_lidiff_split3", None)
    if hit is not None and hit[0] == (x.data_ptr(), x._version, tuple(x.shape), pieces):
        return hit[1]
    require_device(x)
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous() and x.shape[1] % 8 == 0
    out = torch.empty((x.shape[0], pieces, x.shape[1]), dtype=torch.bfloat16 if pieces == 3 else torch.float16, device=x.device)
    # (d_rows: a matrix handed over at its bound -- the rows behind the device-side count are uninitialised and are left alone)
    call("lidiff_split3_rows", ptr(x), x.shape[0], x.shape[1], pieces, ptr(out), ptr(split_status(x.device)) if pieces == 2 else None,
         ptr(d_rows), x.shape[0] // max(1, replicas), stream_ptr())
    try:
        x._lidiff_split3 = ((x.data_ptr(), x._version, tuple(x.shape), pieces), out)
    except AttributeError:
        pass
    return out


class SplitRangeError(RuntimeError):
    """A value beyond fp16's range met the two-piece fp16 convolution: the results since the last split_check() are void."""


_SPLIT_STATUS = {}


def split_status(device) -> torch.Tensor:
    """The device word the two-piece fp16 kernels report values beyond fp16's range in (split_check reads it)."""
    idx = torch.cuda.current_device() if getattr(device, "index", None) is None else device.index
    if idx not in _SPLIT_STATUS:
        _SPLIT_STATUS[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return _SPLIT_STATUS[idx]


def split_check() -> None:
    """Raises if a two-piece fp16 convolution since the last call met a value it cannot hold (one host read; a no-op when the mode was
    never used).  DiffCompletion calls it at the end of a scan."""
    for t in _SPLIT_STATUS.values():
        flags = int(t.item())
        if flags:
            t.zero_()
        if flags & STATUS_F16_RANGE:
            raise SplitRangeError("a feature or weight beyond fp16's range (|x| > 65504, Inf or NaN) reached the two-piece fp16 convolution "
                               "(LIDIFF_SPLIT_PIECES=2): results are void -- run with the default three bf16 pieces")


def packed_weights_split(w: torch.Tensor, pieces: int):
    """(packed weights, out_scale) of spconv_fwd_split3: three bf16 pieces (out_scale 1), or two fp16 pieces of w * 2^k with 2^k
    putting max |w| into [2^11, 2^12) -- every second piece a normal fp16 number, products far inside fp32's range -- and out_scale =
    2^-k.  Cached on the Parameter (the scale costs one host read per weight version)."""
    if pieces == 3:
        return packed_weights_bf16(w, planes=3), 1.0
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    key = (w.data_ptr(), w._version, tuple(w3.shape), w.device)
    hit = getattr(w, "_lidiff_packed_f16x2", None)
    if hit is None or hit[0] != key:
        src = w3.detach().contiguous().float()
        k, c_in, c_out = src.shape
        top = float(src.abs().max())
        if not math.isfinite(top):
            raise RuntimeError("non-finite convolution weights")
        scale = 2.0 ** (11 - math.floor(math.log2(top))) if top > 0.0 else 1.0
        wp = torch.empty(_lib.load().lidiff_spconv_packed_weight_bf16_elems(k, c_in, c_out, 2), dtype=torch.float16, device=w.device)
        call("lidiff_spconv_pack_weights_f16x2", ptr(src), k, c_in, c_out, float(scale), ptr(wp), ptr(split_status(w.device)), stream_ptr())
        hit = (key, wp, 1.0 / scale)
        try:
            setattr(w, "_lidiff_packed_f16x2", hit)
        except AttributeError:
            pass
    return hit[1], hit[2]


def mask_sorted_map(nbr: torch.Tensor):
    """(nbr[:, order], order) with the rows of a neighbour table sorted by their neighbour sets (lidiff_row_mask_keys, stable
    descending sort): the row order spconv_fwd_split3 skips whole 16-row blocks under.  Valid rows first (a table handed over at
    its bound keeps its device-side count).  Cached on the table tensor."""
    hit = getattr(nbr, "_lidiff_mask_sorted", None)
    if hit is None:
        require_device(nbr)
        k, m = nbr.shape
        keys = torch.empty(m, dtype=torch.int32, device=nbr.device)
        call("lidiff_row_mask_keys", ptr(nbr), k, m, ptr(keys), stream_ptr())
        order = torch.sort(keys, descending=True, stable=True).indices
        hit = (nbr.index_select(1, order).contiguous(), order.to(torch.int32))
        nbr._lidiff_mask_sorted = hit
    return hit


def split3_conv_applies(c_in_a: int, c_in_b: int, c_out: int) -> bool:
    return bool(_lib.load().lidiff_spconv_fwd_split3_supported(int(c_in_a), int(c_in_b), int(c_out)))


def spconv_fwd_split3(in_a, w: torch.Tensor, nbr: torch.Tensor | None, m_out: int, in_b=None, scale=None, shift=None,
                      residual=None, relu: bool = False, replicas: int = 1, d_rows: torch.Tensor | None = None,
                      want_planes: bool = False, row_order: torch.Tensor | None = None, pieces: int | None = None,
                      d_in_rows: torch.Tensor | None = None) -> torch.Tensor:
    """spconv_fwd (fp32 in, fp32 out, fp32 accuracy) with the contraction on the bf16 matrix pipe from three-way split operands
    (lidiff_spconv_fwd_split3; include/lidiff_amd.h).  in_a / in_b: fp32 [R * M_in, C] (cut here, the cut cached on the
    tensor) or the bf16 [R * M_in, 3, C] pieces themselves.  want_planes: the result carries its own pieces
    (`_lidiff_split3`, written by the kernel's epilogue) for the next dense convolution.  row_order: int32 permutation of the
    output rows with `nbr` holding its columns in that order (mask_sorted_map); results do not depend on it.  pieces: 3 bf16 (default,
    ops.SPLIT_PIECES) or 2 fp16 pieces per operand."""
    require_device(w, nbr, scale, shift, residual, row_order)
    pieces = SPLIT_PIECES if pieces is None else int(pieces)
    pdt = torch.bfloat16 if pieces == 3 else torch.float16
    if row_order is not None:
        assert row_order.dtype == torch.int32 and row_order.shape == (m_out,) and row_order.is_contiguous() and nbr is not None
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    k, c_in, c_out = w3.shape
    wp, out_scale = packed_weights_split(w, pieces)
    # (for a kernel_size-3 map onto itself the inputs' valid rows are the outputs': d_rows describes both)
    in_rows = d_in_rows if d_in_rows is not None else (d_rows if (nbr is None or in_a.shape[0] == replicas * m_out) else None)
    a3 = in_a if in_a.dtype == pdt else split3_rows(in_a, pieces, in_rows, replicas)
    b3 = None if in_b is None else (in_b if in_b.dtype == pdt else split3_rows(in_b, pieces, in_rows, replicas))
    assert a3.dim() == 3 and a3.shape[1] == pieces and a3.dtype == pdt and a3.is_contiguous() and a3.shape[0] % replicas == 0
    assert b3 is None or (b3.dim() == 3 and b3.shape[1] == pieces and b3.dtype == pdt)
    c_a, c_b = a3.shape[2], 0 if b3 is None else b3.shape[2]
    assert c_a + c_b == c_in, f"channel mismatch {c_a}+{c_b} != {c_in}"
    m_in = a3.shape[0] // replicas
    if nbr is not None:
        assert nbr.shape == (k, m_out) and nbr.dtype == torch.int32 and nbr.is_contiguous()
    if residual is not None:
        residual = residual.contiguous()
        assert residual.shape == (replicas * m_out, c_out)
    out = torch.empty((replicas * m_out, c_out), dtype=torch.float32, device=a3.device)
    out3 = torch.empty((replicas * m_out, pieces, c_out), dtype=pdt, device=a3.device) if want_planes else None
    prof = PROFILER
    timed = prof is not None and prof.wants("split3")
    start = end = None
    if timed:
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _lib.join_pending()
        start.record()
    call("lidiff_spconv_fwd_split3", ptr(a3), c_a, ptr(b3), c_b, ptr(wp), ptr(nbr), k, m_in, m_out, c_out, ptr(out), ptr(out3),
         ptr(scale), ptr(shift), ptr(residual), int(bool(relu)), int(replicas), ptr(d_rows), ptr(row_order), pieces, float(out_scale),
         ptr(split_status(a3.device)) if pieces == 2 else None, stream_ptr())
    if timed:
        end.record()
    if prof is not None:
        prof.launches.append(("split3", start, end, m_in, m_out, c_in, c_out, k, nbr, replicas))
    if out3 is not None:
        out._lidiff_split3 = ((out.data_ptr(), out._version, tuple(out.shape), pieces), out3)
    return out


def rulebook_of(nbr: torch.Tensor):
    """(pairs_in, pairs_out, offset_ptr, n_pairs) of a neighbour table, built once and kept on the table tensor
    (kernel maps are cached per coordinate manager, so every conv on the map shares it in backward)."""
    rb = getattr(nbr, "_lidiff_rulebook", None)
    if rb is None:
        pin, pout, off = rulebook_compact(nbr)
        rb = (pin, pout, off, int(pin.shape[0]))
        nbr._lidiff_rulebook = rb
    return rb


def build_rulebooks(tables) -> None:
    """rulebook_of() for several neighbour tables with ONE host read: every table's counting pass is queued first, the
    totals come back together, then the fill passes (a training step walks ~13 kernel maps in its backward; one read each
    left the GPU idle while the host waited -- profiles/r03_train_idle_gaps.txt)."""
    todo = [t for t in tables if t is not None and getattr(t, "_lidiff_rulebook", None) is None]
    if not todo:
        return
    require_device(*todo)
    offs = []
    for nbr in todo:
        k, m = nbr.shape
        ws = torch.empty(_lib.load().lidiff_rulebook_workspace_bytes(k, m), dtype=torch.uint8, device=nbr.device)
        off = torch.empty(k + 1, dtype=torch.int32, device=nbr.device)
        call("lidiff_rulebook_compact", ptr(nbr), k, m, ptr(off), None, None, ptr(ws), stream_ptr())
        offs.append((off, ws))
    totals = torch.stack([off[-1] for off, _ in offs]).tolist()                      # the one read
    for nbr, (off, ws), total in zip(todo, offs, totals):
        k, m = nbr.shape
        pin = torch.empty(total, dtype=torch.int32, device=nbr.device)
        pout = torch.empty(total, dtype=torch.int32, device=nbr.device)
        call("lidiff_rulebook_compact", ptr(nbr), k, m, ptr(off), ptr(pin), ptr(pout), ptr(ws), stream_ptr())
        nbr._lidiff_rulebook = (pin, pout, off, int(total))


def spconv_bwd_w(in_a, grad_out, nbr, k: int, in_b=None, bf16: bool = False) -> torch.Tensor:
    """Weight gradient of spconv_fwd (training path, models.py:180-217): dW[k] = gather(in)[pairs_k]^T @
    grad_out[pairs_k] by the MFMA kernel lidiff_spconv_bwd_w over the map's rulebook; input channel counts that are not
    multiples of 4 (the 3-channel stem) are zero-padded to the next multiple for the kernel.
    bf16: operands rounded to bf16, fp32 sums (lidiff_spconv_bwd_w_bf16) -- the bf16 training configuration."""
    require_device(in_a, grad_out, nbr, in_b)
    in_a = in_a.contiguous()
    grad_out = grad_out.contiguous()
    c_a, c_b = in_a.shape[1], 0 if in_b is None else in_b.shape[1]
    m_out, c_out = grad_out.shape
    rows16 = in_a.dtype == torch.bfloat16                # bf16 shadow rows of BOTH operands (cast_bf16)
    if rows16:
        assert bf16 and grad_out.dtype == torch.bfloat16 and (in_b is None or in_b.dtype == torch.bfloat16)
        assert c_a % 4 == 0 and c_b % 4 == 0 and c_out % 4 == 0
    if c_a % 4 == 0 and c_b % 4 == 0 and c_out % 4 == 0:
        if in_b is not None:
            in_b = in_b.contiguous()
        pin, pout, off, n_pairs = (None, None, None, m_out) if nbr is None else rulebook_of(nbr)
        ws = None
        if DETERMINISTIC_DW:            # pair slices summed in a fixed order through a workspace (no atomics)
            nws = _lib.load().lidiff_spconv_bwd_w_workspace_floats(c_a + c_b, c_out, k, n_pairs)
            ws = torch.empty(nws, dtype=torch.float32, device=in_a.device) if nws else None
        alloc = torch.empty if ws is not None else torch.zeros
        dw = alloc((k, c_a + c_b, c_out), dtype=torch.float32, device=in_a.device)
        prof = PROFILER
        timed = prof is not None and prof.wants("dw")
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if bf16:
            call("lidiff_spconv_bwd_w_bf16", ptr(in_a), c_a, ptr(in_b), c_b, ptr(grad_out), ptr(pin), ptr(pout), ptr(off),
                 n_pairs, k, in_a.shape[0], m_out, c_out, ptr(dw), ptr(ws), int(rows16), stream_ptr())
        else:
            call("lidiff_spconv_bwd_w", ptr(in_a), c_a, ptr(in_b), c_b, ptr(grad_out), ptr(pin), ptr(pout), ptr(off),
                 n_pairs, k, in_a.shape[0], m_out, c_out, ptr(dw), ptr(ws), stream_ptr())
        if timed:
            ev[1].record()
            prof.dw.append(ev)
        return dw
    # channel counts that are not multiples of 4 (the 3-channel stem, models.py: in_channels = 3): zero-pad the input columns
    # to the next multiple of 4 and take the MFMA kernel -- the padded columns' gradient rows are dropped.  (Rounds 1-2 looped
    # over the offsets with two row gathers and a library GEMM each: 27 small launch triples and a host read per step.)
    x = in_a if in_b is None else torch.cat([in_a, in_b], dim=1)
    c_in = x.shape[1]
    if c_out % 4 == 0:
        pad = (-c_in) % 4
        xp = torch.nn.functional.pad(x, (0, pad)) if pad else x
        return spconv_bwd_w(xp, grad_out, nbr, k, bf16=False)[:, :c_in, :].contiguous()
    if nbr is None:                                   # kernel_size 1: identity map
        return (x.t() @ grad_out).unsqueeze(0)
    pin, pout, off, _ = rulebook_of(nbr)
    off = off.tolist()
    dw = torch.zeros((k, x.shape[1], c_out), dtype=torch.float32, device=x.device)
    for kk in range(k):
        lo, hi = off[kk], off[kk + 1]
        if hi > lo:
            dw[kk] = gather_rows(x, pin[lo:hi].long()).t() @ gather_rows(grad_out, pout[lo:hi].long())
    return dw


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """SparseTensor.slice(field).F (minkunet.py:497,619) and x_part.F[idx] (:418)."""
    require_device(src, idx)
    src = src.contiguous()
    idx = idx.contiguous()
    assert idx.dtype == torch.int64
    n, c = idx.shape[0], src.shape[1]
    dst = torch.empty((n, c), dtype=torch.float32, device=src.device)
    call("lidiff_gather_rows", ptr(src), ptr(idx), n, c, ptr(dst), stream_ptr())
    return dst


_HEAD_WEIGHTS: dict = {}


def slice_head_applies(head) -> bool:
    """`head` is Sequential(Linear(C, 20), LeakyReLU, Linear(20, 3)) on the device with C % 4 == 0 -- LiDiff's `self.last`
    (minkunet.py:390) -- i.e. what lidiff_slice_head evaluates."""
    mods = list(head)
    if len(mods) != 3 or not (isinstance(mods[0], torch.nn.Linear) and isinstance(mods[1], torch.nn.LeakyReLU)
                              and isinstance(mods[2], torch.nn.Linear)):
        return False
    l1, l2 = mods[0], mods[2]
    if l1.bias is None or l2.bias is None or l1.weight.dtype != torch.float32 or not l1.weight.is_cuda:
        return False
    return bool(_lib.load().lidiff_slice_head_supported(l1.in_features, l1.out_features, l2.out_features))


def slice_head(feats: torch.Tensor, inverse: torch.Tensor, head, replicas: int = 1) -> torch.Tensor:
    """head(feats[inverse]) -- SparseTensor.slice(field).F through `self.last` (minkunet.py:390,497) -- in one launch, for the
    `replicas` stacked voxel matrices of a CFG pair ([R * M, C] -> [R * len(inverse), 3]).  Inference only (no autograd)."""
    require_device(feats, inverse)
    l1, act, l2 = list(head)
    key = id(l1)
    ver = (l1.weight._version, l1.weight.data_ptr())
    hit = _HEAD_WEIGHTS.get(key)
    if hit is None or hit[0] != ver:
        hit = _HEAD_WEIGHTS[key] = (ver, l1.weight.detach().t().contiguous())            # [C, hidden]
    w1t = hit[1]
    feats, inverse = feats.contiguous(), inverse.contiguous()
    assert inverse.dtype == torch.int64 and feats.dtype == torch.float32 and feats.shape[0] % replicas == 0
    n, c = inverse.shape[0], feats.shape[1]
    out = torch.empty((replicas * n, l2.out_features), dtype=torch.float32, device=feats.device)
    call("lidiff_slice_head", ptr(feats), ptr(inverse), n, feats.shape[0] // replicas, int(replicas), c, ptr(w1t),
         ptr(l1.bias.detach()), l1.out_features, ptr(l2.weight.detach().contiguous()), ptr(l2.bias.detach()), l2.out_features,
         float(act.negative_slope), ptr(out), stream_ptr())
    return out


def gather_bias_leaky(src: torch.Tensor, idx: torch.Tensor, bias: torch.Tensor, slope: float,
                      out: torch.Tensor | None = None) -> torch.Tensor:
    """leaky_relu(src[idx] + bias, slope) in one pass (conditioning MLP hidden layer, minkunet.py:424-431).
    `out`: optional [len(idx), C] destination (a row slice of a larger buffer)."""
    require_device(src, idx, bias)
    src, idx, bias = src.contiguous(), idx.contiguous(), bias.contiguous().reshape(-1)
    n, c = idx.shape[0], src.shape[1]
    assert bias.shape[0] == c and idx.dtype == torch.int64
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    assert out.shape == (n, c) and out.is_contiguous()
    call("lidiff_gather_bias_leaky", ptr(src), ptr(idx), ptr(bias), n, c, float(slope), ptr(out), stream_ptr())
    return out


def gather_mul_rows(x: torch.Tensor, table: torch.Tensor, idx: torch.Tensor | None, out: torch.Tensor | None = None,
                    d_rows: torch.Tensor | None = None):
    """x * table[idx] in one pass (the conditioning multiply of minkunet.py:431 etc. with its MLP evaluated on the
    part rows).  `out`: optional destination of x's shape (may be a row slice of a larger buffer).  idx None: every row takes
    table row 0 (a broadcast).  d_rows: int32 [1], the valid rows when x's row count is only a bound."""
    require_device(x, table, idx, d_rows)
    x, table = x.contiguous(), table.contiguous()
    n, c = x.shape
    assert table.shape[1] == c
    if idx is not None:
        idx = idx.contiguous()
        assert idx.shape[0] == n and idx.dtype == torch.int64
    if out is None:
        out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    assert out.shape == (n, c) and out.is_contiguous()
    call("lidiff_gather_mul_rows", ptr(x), ptr(table), ptr(idx), n, c, ptr(out), ptr(d_rows), stream_ptr())
    return out


# scatter-adds (the backward of row gathers) run as segment sums over a destination-sorted source list: deterministic, no atomics


def _scatter_csr(idx: torch.Tensor, m: int):
    """(order, ptr) of a gather index: its sources stably sorted by destination row and the CSR over the m destination rows.
    Cached on the index tensor (a step's inverse mapping / match index is scattered through once per consumer layer)."""
    key = (idx.data_ptr(), idx._version, idx.shape[0], m)
    hit = getattr(idx, "_lidiff_csr", None)
    if hit is None or hit[0] != key:
        sorted_idx, order = torch.sort(idx, stable=True)
        ptr_ = torch.searchsorted(sorted_idx, torch.arange(m + 1, device=idx.device, dtype=idx.dtype))
        hit = (key, order.contiguous(), ptr_.contiguous())
        try:
            idx._lidiff_csr = hit
        except AttributeError:
            pass
    return hit[1], hit[2]


def scatter_add_rows(src: torch.Tensor, idx: torch.Tensor, m: int) -> torch.Tensor:
    """dst[idx[i]] += src[i] over the rows i (dst [m, C]): the backward of gather_rows."""
    src = src.contiguous()
    n, c = src.shape
    if n > 0:
        order, ptr_ = _scatter_csr(idx.contiguous(), m)
        dst = torch.empty((m, c), dtype=torch.float32, device=src.device)
        # destinations with > 64 sources (the unconditional training branch: ~180 000 rows gathered from each of 2 part voxels)
        # are summed in chunks by whole workgroups instead of by one thread walking the segment
        work = torch.empty(_lib.load().lidiff_segment_sum_workspace_bytes(n, c), dtype=torch.uint8, device=src.device)
        call("lidiff_segment_sum_rows", ptr(src), ptr(order), ptr(ptr_), m, c, ptr(dst), n, ptr(work), stream_ptr())
        return dst
    return torch.zeros((m, c), dtype=torch.float32, device=src.device)


FPS_COOPERATIVE = True


def _inv_resolution(resolution: float) -> float:
    """1 / resolution as torch's GPU kernels form it for ``x / resolution`` with a host scalar: the reciprocal in the opmath
    type of float32 tensors (0.05 -> exactly 20.0f)."""
    import numpy as np
    return float(np.float32(1.0) / np.float32(resolution))


def points_to_field(points: torch.Tensor, resolution: float, scale_batch_column: bool = True):
    """[B, n, 3] device points (float64 or float32) -> (float32 features [B n, 3], int32 voxel coordinates [B n, 4]) in one
    launch: batched_coordinates + round(x / resolution) + ME's floor (pipeline:68-84; models.py:162-178 with
    scale_batch_column=False)."""
    require_device(points)
    if points.dim() != 3 or points.shape[2] != 3 or points.dtype not in (torch.float32, torch.float64):
        raise ValueError("points must be [B, n, 3] float32 / float64")
    points = points.contiguous()
    n = points.shape[0] * points.shape[1]
    feats = torch.empty((n, 3), dtype=torch.float32, device=points.device)
    coords = torch.empty((n, 4), dtype=torch.int32, device=points.device)
    call("lidiff_points_to_field", ptr(points), int(points.dtype == torch.float64), _inv_resolution(resolution), n,
         max(1, points.shape[1]), int(bool(scale_batch_column)), ptr(feats), ptr(coords), stream_ptr())
    return feats, coords


def cfg_dpm_step(e_cond, e_uncond, w: float, x_t, x_init, plan: dict, noise, resolution: float, scale_batch_column: bool = True):
    """Guidance + DPM-Solver++ update + the next field's points and voxel coordinates as one launch (lidiff_cfg_dpm_step;
    pipeline:148-153,161-164).  plan: DPMSolverMultistepScheduler.step_plan(t).  Returns (x0 [B, n, 3] float64, features
    [B n, 3] float32, coordinates [B n, 4] int32)."""
    import numpy as np
    require_device(e_cond, e_uncond, x_t, x_init)
    b, n_per = x_init.shape[0], x_init.shape[1]
    n = b * n_per
    e_cond, e_uncond, x_t, x_init = (v.contiguous() for v in (e_cond, e_uncond, x_t, x_init))
    if not (e_cond.dtype == e_uncond.dtype == x_t.dtype == torch.float32 and x_init.dtype == torch.float64
            and e_cond.numel() == e_uncond.numel() == x_t.numel() == x_init.numel() == 3 * n):
        raise ValueError("cfg_dpm_step: float32 eps / points [B, n, 3] and float64 x_init [B, n, 3] expected")
    m_prev = plan["m_prev"]
    if m_prev is not None:
        m_prev = m_prev.contiguous()
        assert m_prev.dtype == torch.float64 and m_prev.numel() == 3 * n
    if noise is not None:
        noise = noise.contiguous()
        assert noise.dtype == torch.float64 and noise.numel() == 3 * n
    x0 = torch.empty((b, n_per, 3), dtype=torch.float64, device=x_init.device)
    feats = torch.empty((n, 3), dtype=torch.float32, device=x_init.device)
    coords = torch.empty((n, 4), dtype=torch.int32, device=x_init.device)
    call("lidiff_cfg_dpm_step", ptr(e_cond), ptr(e_uncond), float(w), ptr(x_t), ptr(x_init), ptr(m_prev), ptr(noise),
         float(np.float32(plan["sigma_t"])), 1.0 / plan["alpha_t"], plan["c_sample"], plan["c_m0"], plan["c_d1"], plan["inv_r0"],
         plan["c_noise"], _inv_resolution(resolution), n, max(1, n_per), int(bool(scale_batch_column)), ptr(x0), ptr(feats),
         ptr(coords), stream_ptr())
    return x0, feats, coords


def farthest_point_sample(points: torch.Tensor, n_samples: int) -> torch.Tensor:
    """open3d farthest_point_down_sample stand-in of preprocess_scan (pipeline:97-99): indices of n_samples
    points, greedy from index 0, float64 squared distances, first maximum wins."""
    require_device(points)
    pts = points.contiguous().double()
    n = pts.shape[0]
    if n_samples >= n:
        return torch.arange(n, device=pts.device)
    sel = torch.empty(n_samples, dtype=torch.int64, device=pts.device)
    ws = torch.empty(_lib.load().lidiff_fps_workspace_bytes(n), dtype=torch.uint8, device=pts.device)
    if FPS_COOPERATIVE and _lib.load().lidiff_fps_coop_supported(n):
        # one persistent cooperative launch (device-wide barrier per selection).  Its status word is read back -- the caller
        # indexes with the result on the host side of the step anyway.  Only a barrier time-out (reported, not silent)
        # falls back to the launch-per-selection kernel below (same indices); every other failure raises.
        status = torch.empty(1, dtype=torch.int32, device=pts.device)
        call("lidiff_fps_coop", ptr(pts), n, int(n_samples), ptr(sel), ptr(ws), ptr(status), stream_ptr())
        if int(status.item()) == 0:
            return sel
        import warnings
        warnings.warn("lidiff_fps_coop: device-wide barrier timed out; re-running the selection with lidiff_fps")
    call("lidiff_fps", ptr(pts), n, int(n_samples), ptr(sel), ptr(ws), stream_ptr())
    return sel


# nn_dist goes through the uniform grid (lidiff_nn_dist_grid: same results, bit for bit) from this many point pairs on
NN_GRID_MIN_PAIRS = int(2e8)
NN_GRID_CELL = 0.5          # metres (LiDiff's clouds): ~25 surface points per cell at the scans' density; results do not depend on it


def nn_dist(a: torch.Tensor, b: torch.Tensor, grid: bool | None = None, cell: float | None = None):
    """Squared distance and row index of the nearest point of b [M,3] for every point of a [N,3] (float32 or
    float64, lowest index on ties): open3d compute_point_cloud_distance (utils/metrics.py:68,128-129) and the
    K=1 search of pytorch3d chamfer_distance (models_refine.py:72).  Returns (d2 [N], idx int64 [N]).
    grid: search through a uniform grid over b (cells of edge `cell`) instead of scanning all of b -- exact, identical results;
    None = from NN_GRID_MIN_PAIRS point pairs on."""
    require_device(a, b)
    if a.dtype != b.dtype or a.dtype not in (torch.float32, torch.float64):
        raise TypeError("nn_dist needs two float32 or two float64 clouds")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 3 or b.shape[1] != 3:
        raise ValueError("nn_dist needs [N,3] and [M,3] clouds")
    a = a.detach().contiguous()
    b = b.detach().contiguous()
    n, m, eb = a.shape[0], b.shape[0], a.element_size()
    d2 = torch.empty(n, dtype=a.dtype, device=a.device)
    idx = torch.empty(n, dtype=torch.int64, device=a.device)
    if grid is None:
        grid = n * m >= NN_GRID_MIN_PAIRS
    if grid and m >= 1 and n >= 1:
        ws = torch.empty(_lib.load().lidiff_nn_dist_grid_workspace_bytes(n, m, eb), dtype=torch.uint8, device=a.device)
        call("lidiff_nn_dist_grid", ptr(a), n, ptr(b), m, eb, float(cell or NN_GRID_CELL), ptr(d2), ptr(idx), ptr(ws), stream_ptr())
        return d2, idx
    ws = torch.empty(_lib.load().lidiff_nn_dist_workspace_bytes(n, m, eb), dtype=torch.uint8, device=a.device)
    call("lidiff_nn_dist", ptr(a), n, ptr(b), m, eb, ptr(d2), ptr(idx), ptr(ws), stream_ptr())
    return d2, idx


def nn_match(full_c: torch.Tensor, part_c: torch.Tensor, by_batch: bool = False) -> torch.Tensor:
    """MinkUNetDiff.match_part_to_full index part (minkunet.py:403-416): exhaustive scan of the part rows.
    by_batch: batches of several scans -- every row against its own batch element's part rows first, the unrestricted scan only
    if a check on the device finds a row whose winner that does not settle (lidiff_nn_match d_gate): the same indices."""
    require_device(full_c, part_c)
    full_c = full_c.contiguous()
    part_c = part_c.contiguous()
    assert full_c.dtype == torch.int32 and part_c.dtype == torch.int32
    max_coord = full_c.max().to(torch.int32).reshape(1)
    idx = torch.empty(full_c.shape[0], dtype=torch.int64, device=full_c.device)
    gate = torch.empty(1, dtype=torch.int32, device=full_c.device) if by_batch else None
    call("lidiff_nn_match", ptr(full_c), full_c.shape[0], ptr(part_c), part_c.shape[0], ptr(max_coord), ptr(idx), ptr(gate),
         stream_ptr())
    return idx


def nn_match_dev(full_bound: torch.Tensor, d_count: torch.Tensor, part_c: torch.Tensor) -> torch.Tensor:
    """nn_match for a map whose row count still lives on the device (build_pyramid before its host read): full_bound
    [n_bound, 4] int32 of which the first d_count[0] rows are valid.  Returns idx [n_bound] int64 (valid rows written)."""
    require_device(full_bound, d_count, part_c)
    part_c = part_c.contiguous()
    assert full_bound.dtype == torch.int32 and part_c.dtype == torch.int32 and d_count.dtype == torch.int32 and full_bound.is_contiguous()
    max_coord = torch.empty(1, dtype=torch.int32, device=full_bound.device)
    idx = torch.empty(full_bound.shape[0], dtype=torch.int64, device=full_bound.device)
    call("lidiff_nn_match_dev", ptr(full_bound), full_bound.shape[0], ptr(d_count), ptr(part_c), part_c.shape[0], ptr(max_coord),
         ptr(idx), stream_ptr())
    return idx


def argmin_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """idx[i] = argmin_j |a_i - b_j|^2 (fp32, lowest j on ties) for rows of dimension <= 4: the pykeops expression of
    minkunet.py:412-416 evaluated by the HIP brute-force kernel (lidiff_argmin_rows_f32)."""
    require_device(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.shape[1] > 4:
        raise ValueError("argmin_rows needs [N,D] and [M,D] rows with D <= 4")
    pad = lambda t: torch.nn.functional.pad(t.float(), (0, 4 - t.shape[1])).contiguous()
    a4, b4 = pad(a), pad(b)
    idx = torch.empty(a4.shape[0], dtype=torch.int64, device=a.device)
    call("lidiff_argmin_rows_f32", ptr(a4), a4.shape[0], ptr(b4), b4.shape[0], ptr(idx), stream_ptr())
    return idx


class TailMap:
    """The non-centre pairs of a kernel_size-3 / stride-1 kernel map, laid out for the two-pass convolution of
    low-density maps (include/lidiff_amd.h, lidiff_spconv_fwd `tail`):
      nbr [27, P]   kernel map of the P pairs as output rows: row p (pairs sorted by offset, then output row) has its input
                    row at its own offset and -1 elsewhere -- lidiff_spconv_fwd over it multiplies every pair with its
                    W[k], 128 pairs of (mostly) ONE offset per tile;
      ptr [M + 1], idx [P]   CSR over the map's output rows: the pairs landing on output row o, in ascending offset.
    Built once per coordinate map from its neighbour table (shared by every convolution on the map)."""

    bounded = False

    def __init__(self, nbr: torch.Tensor | None, phase1=None, bounded=None):
        if phase1 is not None:          # build_pyramid(): counts taken with the row count on the device, size already read
            self._from_phase1(*phase1)
            return
        if bounded is not None:         # ... and never read: the pair count stays on the device, `n` is its bound
            nbr_bound, m_bound, d_m, ws, off, row_ptr, n_bound, status = bounded
            self.bounded, self.n, self.off, self.ptr = True, int(n_bound), off, row_ptr
            self.nbr = self.idx = self.pair_in = None
            self._pending = (nbr_bound, m_bound, d_m, ws, row_ptr, status)
            return
        require_device(nbr)
        k, m = nbr.shape
        assert k == 27 and nbr.dtype == torch.int32 and nbr.is_contiguous()
        dev = nbr.device
        ws = torch.empty(_lib.load().lidiff_tail_map_workspace_bytes(k, m), dtype=torch.uint8, device=dev)
        off = torch.empty(k + 1, dtype=torch.int32, device=dev)
        self.ptr = torch.empty(m + 1, dtype=torch.int32, device=dev)
        call("lidiff_tail_map", ptr(nbr), k, m, 13, ptr(off), ptr(self.ptr), 0, None, None, ptr(ws), stream_ptr())
        self.n = int(off[-1].item())                                     # the one host read of this map
        self.nbr = self.idx = self.pair_in = None
        self.off = off                                                   # [28] pair ranges of the offsets (13: empty)
        if self.n:
            self.nbr = torch.empty((k, self.n), dtype=torch.int32, device=dev)
            self.idx = torch.empty(self.n, dtype=torch.int32, device=dev)
            call("lidiff_tail_map", ptr(nbr), k, m, 13, ptr(off), ptr(self.ptr), self.n, ptr(self.nbr), ptr(self.idx),
                 ptr(ws), stream_ptr())
            self.pair_in = self.nbr.amax(0)                              # the pair list form of the same map (one entry per column)

    @classmethod
    def on_device(cls, nbr_bound: torch.Tensor, d_m: torch.Tensor, status: torch.Tensor, n_bound: int | None = None):
        """The tail map of a kernel map whose row count lives on the device (nbr_bound [27, m_bound], columns behind the count
        -1): counting phase now, the pair lists (bounded by n_bound pairs, default TAIL_PAIR_BOUND per row of the bound) when
        fill() is called; no host read at all."""
        require_device(nbr_bound, d_m, status)
        k, m_bound = nbr_bound.shape
        assert k == 27 and nbr_bound.dtype == torch.int32 and nbr_bound.is_contiguous()
        dev = nbr_bound.device
        ws = torch.empty(_lib.load().lidiff_tail_map_workspace_bytes(k, m_bound), dtype=torch.uint8, device=dev)
        off = torch.empty(k + 1, dtype=torch.int32, device=dev)
        row_ptr = torch.empty(m_bound + 1, dtype=torch.int32, device=dev)
        call("lidiff_tail_map_dev", ptr(nbr_bound), k, m_bound, ptr(d_m), 13, ptr(off), ptr(row_ptr), 0, None, None, ptr(ws), stream_ptr())
        return cls(None, bounded=(nbr_bound, m_bound, d_m, ws, off, row_ptr,
                                  max(1, int(TAIL_PAIR_BOUND * m_bound)) if n_bound is None else n_bound, status))

    def _from_phase1(self, nbr_bound, m_bound, d_m, ws, off, row_ptr, n, m):
        """Phase 2 over the table of pitch m_bound whose phase 1 ran in build_pyramid(); filled when first asked for."""
        self.n, self.off, self.ptr = int(n), off, row_ptr[:m + 1]
        self.nbr = self.idx = self.pair_in = None
        self._pending = (nbr_bound, m_bound, d_m, ws, row_ptr) if self.n else None

    def fill(self):
        """The pair arrays (nbr / idx / pair_in) of a map that came out of build_pyramid(): no host read (n is known)."""
        pend = getattr(self, "_pending", None)
        if pend is not None and self.bounded:
            nbr_bound, m_bound, d_m, ws, row_ptr, status = pend
            dev = nbr_bound.device
            self.pair_in = torch.empty(self.n, dtype=torch.int32, device=dev)
            self.idx = torch.empty(self.n, dtype=torch.int32, device=dev)
            call("lidiff_tail_map_fill_bounded", ptr(nbr_bound), 27, m_bound, ptr(d_m), 13, ptr(self.off), ptr(row_ptr), self.n,
                 ptr(self.pair_in), ptr(self.idx), ptr(status), ptr(ws), stream_ptr())
            self._pending = None
        elif pend is not None:
            nbr_bound, m_bound, d_m, ws, row_ptr = pend
            dev = nbr_bound.device
            self.nbr = torch.empty((27, self.n), dtype=torch.int32, device=dev)
            self.idx = torch.empty(self.n, dtype=torch.int32, device=dev)
            call("lidiff_tail_map_dev", ptr(nbr_bound), 27, m_bound, ptr(d_m), 13, ptr(self.off), ptr(row_ptr), self.n,
                 ptr(self.nbr), ptr(self.idx), ptr(ws), stream_ptr())
            self.pair_in = self.nbr.amax(0)
            self._pending = None
        return self


# (The centre pass keeps the narrow 3 x 2 / 4 x 2 wave grids of the hinted maps: the 96-column tile of the dense family, which
# would read every input row once instead of twice, measured 8 % slower -- profiles/r03_hint_sweep.txt.)
def spconv_centre_tail(in_a, w, tmap: TailMap, m_out, **kw):
    """A kernel_size-3 / stride-1 convolution on a low-density map as two launches: the pairs of the 26 non-centre
    offsets multiplied offset by offset (weight stationary) into one row per pair, then the centre offset as a dense pass
    over contiguous rows that adds those rows through the map's CSR in its epilogue.  Same arguments as spconv_fwd."""
    replicas = kw.get("replicas", 1)
    tail_hint = kw.pop("tail_hint", None)           # the pair count the host believes (bounded maps: for the profiler only)
    tail = None
    if tmap.n > 0:
        in_b = kw.get("in_b")
        if pairs_kernel_applies(in_a.shape[1], 0 if in_b is None else in_b.shape[1], w.shape[-1]):
            rows = spconv_fwd_pairs(in_a, w, tmap.pair_in, None, tmap.off, tmap.n, in_b=in_b, replicas=replicas,
                                    rows_hint=tail_hint, in_rows_hint=kw.get("in_rows_hint"))
        else:
            assert not tmap.bounded, "a bounded tail map is a pair list: shapes outside lidiff_spconv_fwd_pairs take the one-launch kernel"
            rows = spconv_fwd(in_a, w, tmap.nbr, tmap.n, in_b=in_b, replicas=replicas)
        tail = (rows, tmap.ptr, tmap.idx)
    return spconv_fwd(in_a, w, None, m_out, tail=tail, offset=13, **kw)
