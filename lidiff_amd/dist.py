"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Inference (tools/diff_completion_pipeline.py:196-201 processes scans serially; each scan's
T-step trajectory is independent): scans / noise seeds are sharded round-robin over ranks, NO
data-path collective.

Training (train.py:88-101, ``Trainer(accelerator='ddp')`` = DistributedDataParallel over NCCL
plus ``MinkowskiSyncBatchNorm.convert_sync_batchnorm``): gradients are averaged with ONE
flattened all-reduce per bucket after backward.  xGMI is point-to-point (7 links per GPU), so a
few large buckets beat many small ones: the default 64 MiB bucket sends LiDiff's 130.7 MB of
fp32 gradients in three collectives (bf16 transport halves the bytes).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank).  World size 1 needs no
    process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_items(n_items: int, rank: int, world: int) -> list[int]:
    """Round-robin assignment of independent scans / seeds to ranks (scan i -> rank i mod world)."""
    return list(range(rank, n_items, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    """MAX-reduce a host scalar (the bench's step time) over ranks."""
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


class _Bucket:
    __slots__ = ("params", "views", "flat", "wire", "fired", "ready", "work")


class GradAllReducer:
    """Data-parallel gradient averaging with flat buckets (the exchange step of train.py:100, DistributedDataParallel's job).

    Every bucket owns ONE persistent flat buffer.  attach(): the parameters' ``.grad`` ARE views into it (autograd accumulates in
    place), so a bucket is all-reduced where it lies -- no gather into a temporary, no scatter back (round 4 did both with
    ``torch.cat``: 2 x 130 MB of extra traffic per step).  overlap: a post-accumulate hook per parameter launches a bucket's
    all-reduce as soon as backward has produced its last gradient, in bucket order (the same order on every rank, whatever order
    the hooks fire in), while backward is still working on the earlier layers; all_reduce() after backward launches what is left,
    waits, and divides by the world size.  Without attach() the gradients are copied into / out of the buffers (one pass each).
    transport_dtype (e.g. bfloat16): the wire format, through a persistent staging buffer; sums accumulate in that format."""

    def __init__(self, params, bucket_bytes: int = 64 << 20, transport_dtype: torch.dtype | None = None, attach: bool = False,
                 overlap: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.transport_dtype = transport_dtype
        self.buckets: list[_Bucket] = []
        self._bucket_of = {}
        self._next = 0
        groups, cur, size = [], [], 0
        for p in reversed(self.params):          # backward produces the last layers' grads first
            nbytes = p.numel() * p.element_size()
            if cur and (size + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            groups.append(cur)
        for ps in groups:
            b = _Bucket()
            b.params = ps
            b.flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            b.views, off = [], 0
            for p in ps:
                b.views.append(b.flat[off:off + p.numel()].view(p.shape))
                off += p.numel()
                self._bucket_of[p] = b
            b.wire = torch.empty_like(b.flat, dtype=transport_dtype) if transport_dtype else None
            b.fired, b.ready, b.work = 0, False, None
            self.buckets.append(b)
        self.attached = False
        if attach:
            self.attach()
        self._hooks = []
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def close(self):
        """Remove the post-accumulate hooks and let go of the flat buffers: a second reducer on the same module must not find
        this one's hooks still copying gradients and launching all-reduces nobody waits for.  Gradients that are
        views into the buffers are detached copies afterwards."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.attached:
            with torch.no_grad():
                for b in self.buckets:
                    for p, v in zip(b.params, b.views):
                        if p.grad is v:
                            p.grad = v.clone()
            self.attached = False
        self.buckets, self._bucket_of = [], {}

    def __del__(self):
        try:
            for h in self._hooks:
                h.remove()
        except Exception:
            pass

    @staticmethod
    def _active() -> bool:
        return dist.is_initialized() and dist.get_world_size() > 1

    @torch.no_grad()
    def attach(self):
        """Make every parameter's .grad a view into its bucket's buffer (zeros now)."""
        for b in self.buckets:
            b.flat.zero_()
            for p, v in zip(b.params, b.views):
                p.grad = v
        self.attached = True

    @torch.no_grad()
    def zero_grad(self):
        """optimizer.zero_grad() for attached gradients: one fill per bucket; the views stay in place."""
        for b in self.buckets:
            b.flat.zero_()
            for p, v in zip(b.params, b.views):
                if p.grad is not v:
                    p.grad = v

    def _on_grad(self, p):
        if not self._active():
            return
        b = self._bucket_of[p]
        b.fired += 1
        if b.fired >= len(b.params):
            b.ready = True
            while self._next < len(self.buckets) and self.buckets[self._next].ready:
                self._launch(self.buckets[self._next])
                self._next += 1

    @torch.no_grad()
    def _launch(self, b):
        for p, v in zip(b.params, b.views):              # (gradients that do not live in the buffer: copied in)
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        wire = b.flat
        if b.wire is not None:
            wire = b.wire.copy_(b.flat)
        b.work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)

    @torch.no_grad()
    def all_reduce(self):
        if not self._active():
            return
        world = dist.get_world_size()
        while self._next < len(self.buckets):            # whatever backward's hooks have not launched (all of it without overlap)
            self._launch(self.buckets[self._next])
            self._next += 1
        for b in self.buckets:
            b.work.wait()
            if b.wire is not None:
                b.flat.copy_(b.wire)
            b.flat.div_(world)
            for p, v in zip(b.params, b.views):
                if p.grad is None:
                    p.grad = v.clone() if not self.attached else v
                elif p.grad.data_ptr() != v.data_ptr():
                    p.grad.copy_(v)
            b.fired, b.ready, b.work = 0, False, None
        self._next = 0


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Rank 0's weights and buffers to everyone (what DDP does at construction).  A collective writes its tensor
    behind autograd's back (neither ``dist.broadcast(t)`` nor a write through ``t.data`` bumps ``t._version``, which
    keys the packed-weight and folded-BatchNorm caches), so the received values are copied in with ``copy_`` -- an
    ordinary in-place op -- and the derived caches are dropped explicitly on top of that."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            buf = t.detach().clone()
            dist.broadcast(buf, src=src)
            t.copy_(buf)
    from .ops import invalidate_caches
    invalidate_caches(module)
