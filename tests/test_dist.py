"""world_size-2 gloo tests (CPU) of the N>1 path: scan sharding (no collective on the data path),
the bench's max-over-ranks timing reduction, and the data-parallel gradient all-reduce."""
import os
import socket

import torch
import torch.distributed as tdist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from lidiff_amd import dist as ldist
    r, w, _ = ldist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    out = {"shard": ldist.shard_items(7, r, w)}
    out["max"] = ldist.max_over_ranks(1.0 + rank)
    out["sum"] = ldist.sum_over_ranks(10.0 * (rank + 1))
    # data-parallel step: same init everywhere after broadcast, different data per rank
    torch.manual_seed(100 + rank)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 2))
    model(torch.zeros(1, 6))                                       # a forward BEFORE the broadcast ...
    model[0].weight._lidiff_packed = ("stale",)                     # ... leaves derived caches behind
    model[1]._affine_cache = ("stale",)
    v0 = model[0].weight._version
    ldist.broadcast_parameters(model)
    out["caches_dropped"] = not hasattr(model[0].weight, "_lidiff_packed") and not hasattr(model[1], "_affine_cache")
    out["version_bumped"] = model[0].weight._version > v0
    out["w0"] = model[0].weight.detach().clone()
    x = torch.randn(4, 6)
    model(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in model.parameters()]
    ldist.GradAllReducer(model.parameters(), bucket_bytes=64).all_reduce()      # tiny buckets: several collectives
    out["avg"] = [p.grad.clone() for p in model.parameters()]
    gathered = [None] * world
    tdist.all_gather_object(gathered, local)
    out["want"] = [sum(g[i] for g in gathered) / world for i in range(len(local))]
    # bf16 transport path
    model.zero_grad()
    model(x).pow(2).sum().backward()
    ldist.GradAllReducer(model.parameters(), transport_dtype=torch.bfloat16).all_reduce()
    out["avg_bf16"] = [p.grad.clone() for p in model.parameters()]
    # attached + overlapped (what train_loop uses): gradients ARE views into the flat buckets, every bucket is all-reduced in
    # place from a post-accumulate hook as soon as backward has filled it -- same averages, bit for bit, and the views survive
    red = ldist.GradAllReducer(model.parameters(), bucket_bytes=64, attach=True, overlap=True)
    for _ in range(2):                                             # twice: the per-step state is reset
        red.zero_grad()
        model(x).pow(2).sum().backward()
        launched_in_backward = red._next
        red.all_reduce()
    out["avg_overlap"] = [p.grad.clone() for p in model.parameters()]
    out["overlap_views"] = all(p.grad.data_ptr() == v.data_ptr() for b in red.buckets for p, v in zip(b.params, b.views))
    out["overlap_launched_in_backward"] = launched_in_backward
    out["overlap_buckets"] = len(red.buckets)
    # close(): the hooks go (a second reducer on the module must not find the first one's still launching collectives --
    # ADVICE r5), the gradients survive as plain tensors, and a fresh overlapped reducer gives the same averages again
    red.close()
    out["hooks_after_close"] = sum(len(getattr(p, "_post_accumulate_grad_hooks", None) or {}) for p in model.parameters())
    out["grads_after_close"] = all(torch.equal(p.grad, g) for p, g in zip(model.parameters(), out["avg_overlap"]))
    red2 = ldist.GradAllReducer(model.parameters(), bucket_bytes=64, attach=True, overlap=True)
    red2.zero_grad()
    model(x).pow(2).sum().backward()
    red2.all_reduce()
    out["avg_overlap_second"] = [p.grad.clone() for p in model.parameters()]
    out["hooks_second"] = sum(len(getattr(p, "_post_accumulate_grad_hooks", None) or {}) for p in model.parameters())
    red2.close()
    # train_loop: rank r takes batch (step * world + r) mod len -- disjoint data per rank (DistributedSampler), the
    # LR follows Lightning's {'interval': 'epoch', 'frequency': 5} (models.py:340-344), losses are read back lazily
    from lidiff_amd.diffusion import train_loop

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3, 1)
            self.seen = []

        def training_step(self, batch, idx):
            self.seen.append(int(batch["id"]))
            return self.lin(batch["x"]).pow(2).mean()

        def configure_optimizers(self):
            opt = torch.optim.SGD(self.parameters(), lr=1e-3)
            return opt, {"scheduler": torch.optim.lr_scheduler.ExponentialLR(opt, 0.5), "interval": "epoch", "frequency": 5}

    # SyncBatchNorm wiring (train.py:90) on the host side: the conversion gives every MinkowskiBatchNorm an ops.SyncBatchNorm1d
    # child whose process group is the 2-rank world (the statistics themselves run on the HIP kernels: GPU tests); eval mode is
    # the local nn.BatchNorm1d arithmetic on any device
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    holder = torch.nn.Sequential(ME.MinkowskiBatchNorm(8))
    ME.MinkowskiSyncBatchNorm.convert_sync_batchnorm(holder)
    bn = holder[0].bn
    out["sync_bn"] = [type(bn) is ops.SyncBatchNorm1d, bn.group() is not None and tdist.get_world_size(bn.group()) == 2]
    bn.eval()
    xs = torch.randn(5, 8)
    out["sync_bn_eval_equal"] = bool(torch.equal(bn(xs), torch.nn.functional.batch_norm(xs, bn.running_mean, bn.running_var, bn.weight,
                                                                                         bn.bias, False, 0.1, bn.eps)))
    toy = Toy()
    data = [{"id": i, "x": torch.full((2, 3), float(i))} for i in range(8)]
    losses = train_loop(toy, data, steps=40, sync_bn=False, log_every=16)     # 8 batches / 2 ranks = 4 steps per epoch
    out["seen"] = toy.seen[:6]
    out["n_losses"] = len(losses)
    out["toy_w"] = toy.lin.weight.detach().clone()
    ldist.barrier()
    plain = lambda v: v.tolist() if isinstance(v, torch.Tensor) else ([plain(e) for e in v] if isinstance(v, list) else v)
    result_q.put((rank, {k: plain(v) for k, v in out.items()}))   # plain lists: no shared-memory handles
    tdist.destroy_process_group()


def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0]["shard"] == [0, 2, 4, 6] and results[1]["shard"] == [1, 3, 5]
    assert sorted(results[0]["shard"] + results[1]["shard"]) == list(range(7))
    for r in range(world):
        assert results[r]["max"] == 2.0 and results[r]["sum"] == 30.0
        assert results[r]["w0"] == results[0]["w0"]
        assert results[r]["caches_dropped"] and results[r]["version_bumped"]
        assert results[r]["sync_bn"] == [True, True] and results[r]["sync_bn_eval_equal"]
        assert results[r]["seen"] == [(s * 2 + r) % 8 for s in range(6)] and results[r]["n_losses"] == 40
        assert results[r]["toy_w"] == results[0]["toy_w"]                    # averaged gradients: ranks stay in step
        for got, want, got16 in zip(results[r]["avg"], results[r]["want"], results[r]["avg_bf16"]):
            assert torch.allclose(torch.tensor(got), torch.tensor(want), atol=1e-6)
            assert torch.allclose(torch.tensor(got16), torch.tensor(want), rtol=2e-2, atol=2e-2)
    assert results[0]["avg"] == results[1]["avg"]
    for r in (0, 1):
        assert results[r]["avg_overlap"] == results[r]["avg"] and results[r]["overlap_views"]
        assert results[r]["overlap_buckets"] > 1 and results[r]["overlap_launched_in_backward"] == results[r]["overlap_buckets"]
        assert results[r]["hooks_after_close"] == 0 and results[r]["grads_after_close"]
        assert results[r]["avg_overlap_second"] == results[r]["avg"] and results[r]["hooks_second"] == 4


def test_single_process_is_a_noop():
    from lidiff_amd import dist as ldist
    assert ldist.shard_items(3, 0, 1) == [0, 1, 2]
    assert ldist.max_over_ranks(3.5) == 3.5
    m = torch.nn.Linear(2, 2)
    m(torch.ones(1, 2)).sum().backward()
    g = m.weight.grad.clone()
    ldist.GradAllReducer(m.parameters()).all_reduce()
    assert torch.equal(m.weight.grad, g)


def test_lr_schedule_follows_the_reference_interval():
    """models.py:337-346: ExponentialLR(0.5) stepped every 5th epoch -- not per iteration."""
    from lidiff_amd.diffusion import DiffusionPoints, train_loop
    opt, sched = DiffusionPoints.configure_optimizers(
        type("M", (), {"parameters": lambda self: [torch.nn.Parameter(torch.zeros(1))],
                       "hparams": {"train": {"lr": 1e-4}}})())
    assert sched["interval"] == "epoch" and sched["frequency"] == 5 and sched["scheduler"].gamma == 0.5

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(2, 1)
            self.lrs = []

        def training_step(self, batch, idx):
            self.lrs.append(self.opt.param_groups[0]["lr"])
            return self.lin(batch).pow(2).mean()

        def configure_optimizers(self):
            self.opt = torch.optim.Adam(self.parameters(), lr=1e-4)
            return self.opt, {"scheduler": torch.optim.lr_scheduler.ExponentialLR(self.opt, 0.5), "interval": "epoch",
                              "frequency": 5}

    toy = Toy()
    train_loop(toy, [torch.ones(1, 2)] * 3, steps=3 * 11)                    # 11 epochs of 3 steps
    assert toy.lrs[:15] == [1e-4] * 15                                       # epochs 1-5 untouched
    assert toy.lrs[15] == 0.5e-4 and toy.lrs[29] == 0.5e-4 and toy.lrs[30] == 0.25e-4
