"""training_step gradients vs the oracle: table of the worst parameters (debug aid for tests/test_gpu_network.py)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import small_scene
from oracle import minkunet_cpu as net
import lidiff_amd.MinkowskiEngine as ME
from lidiff_amd.diffusion import DiffusionPoints
dev = torch.device("cuda:0")
torch.manual_seed(3)
mod = DiffusionPoints(device=dev)
for m in mod.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        m.weight.data.uniform_(0.8, 1.2); m.bias.data.normal_(0, 0.1)
mod.train()
sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
for k, v in sd.items(): v.requires_grad_(v.is_floating_point() and "running" not in k)
scan, _ = small_scene(seed=9, n=600)
full = torch.from_numpy(np.stack([scan, scan[::-1].copy()])); part = full[:, :60].contiguous()
g = torch.Generator().manual_seed(1)
noise = torch.randn(full.shape, generator=g); t = torch.tensor([700, 30])
def cpu_rounded(points, mean=None, std=None):
    cpu = net.points_to_field(points.detach().cpu().float(), divide_batch_col=False)
    return ME.TensorField(features=points.reshape(-1, 3).float().to(dev), coordinates=cpu.coords_f.to(dev), device=dev)
mod.points_to_tensor = cpu_rounded
names = [k for k, v in sd.items() if v.requires_grad]
for drop in (False, True):
    loss_o, _ = net.training_loss(sd, full, part, noise, t, drop_condition=drop)
    grads_o = torch.autograd.grad(loss_o, [sd[k] for k in names], allow_unused=True)
    for rep in range(2):
        mod.zero_grad(set_to_none=True)
        loss = mod.training_step({"pcd_full": full, "pcd_part": part}, noise=noise, t=t, drop=drop)
        loss.backward()
        params = dict(mod.named_parameters())
        rows = []
        for k, go in zip(names, grads_o):
            gd = params[k].grad
            if go is None or gd is None or (drop and k.startswith("partial_enc.")): continue
            gd = gd.detach().cpu(); n = float(go.norm())
            if n < 1e-7 or not np.isfinite(n): continue
            rows.append(((float((gd - go).norm()) / n), abs(float(gd.norm()) - n) / n, n, k))
        rows.sort(reverse=True)
        print(f"drop={drop} rep={rep} loss {float(loss.detach()):.6f} oracle {float(loss_o.detach()):.6f}; worst by |g-go|/|go|:")
        for r in rows[:6]: print("   rel %.2e  normrel %.2e  norm %.3e  %s" % r)
        print("   worst normrel:", "%.2e %s" % max((r[1], r[3]) for r in rows))

print("---- run-to-run variation of the backward (same inputs, 3 runs): max |g_i - g_0| / |g_0| per parameter, module order")
ref = None
var = {}
for rep in range(3):
    mod.zero_grad(set_to_none=True)
    loss = mod.training_step({"pcd_full": full, "pcd_part": part}, noise=noise, t=t, drop=False)
    loss.backward()
    cur = {k: p.grad.detach().clone() for k, p in mod.named_parameters() if p.grad is not None}
    if ref is None:
        ref = cur
    else:
        for k in cur:
            d = float((cur[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-20))
            var[k] = max(var.get(k, 0.0), d)
for k, p in mod.named_parameters():
    if k in var and (k.endswith("kernel") or k.endswith("bn.bias") or "last" in k or "temp.0.weight" in k):
        print("   %.2e  %s" % (var[k], k))
