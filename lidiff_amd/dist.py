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


class GradAllReducer:
    """Data-parallel gradient averaging with flat buckets (the exchange step of train.py:100)."""

    def __init__(self, params, bucket_bytes: int = 64 << 20, transport_dtype: torch.dtype | None = None):
        self.params = [p for p in params if p.requires_grad]
        self.transport_dtype = transport_dtype
        self.buckets: list[list[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(self.params):          # backward produces the last layers' grads first
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)

    @torch.no_grad()
    def all_reduce(self):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        pending = []
        for bucket in self.buckets:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1) for g in grads])
            wire = flat.to(self.transport_dtype) if self.transport_dtype else flat
            work = dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)
            pending.append((work, wire, bucket, grads))
        for work, wire, bucket, grads in pending:
            work.wait()
            flat = wire.to(grads[0].dtype) / world
            off = 0
            for p, g in zip(bucket, grads):
                n = g.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].reshape(p.shape).clone()
                else:
                    p.grad.copy_(flat[off:off + n].reshape(p.shape))
                off += n


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
