"""Debug aid: run the same closed loop read-free and as its exact-size twin (hint_lag) and report the FIRST fused operator whose
valid output rows differ (python tools/debug/readfree_diff.py [n_base] [steps] [sigma])."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import build_seeded_models  # noqa: E402
from lidiff_amd import minkunet as mn, ops  # noqa: E402
from lidiff_amd.pipeline import DiffCompletion  # noqa: E402

n_base, steps, sigma = int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 3, \
    float(sys.argv[3]) if len(sys.argv) > 3 else 0.6
dev = torch.device("cuda:0")
enc, unet, refine = (m.to(dev).eval() for m in build_seeded_models(42))
fps = np.load(os.path.join(ROOT, "tests", "golden", "scan_000123_fps18000.npy"))
scan = torch.from_numpy(np.tile(fps[:n_base], (10, 1))).double()[None].to(dev)
g = torch.Generator(device="cpu").manual_seed(11)
x0 = (scan.cpu() + sigma * torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(dev)
zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(dev) for _ in range(steps)]

log = []
orig_cba, orig_gmr = mn.conv_bn_act, ops.gather_mul_rows


def valid(f, mgr, ts, reps):
    cnt = mgr.count(ts)
    m = f.shape[0] // reps
    v = m if cnt is None else int(cnt.item())
    return torch.cat([f[r * m:r * m + v] for r in range(reps)]).clone()


def cba(conv, bn, x, relu, residual=None, extra=None):
    out = orig_cba(conv, bn, x, relu, residual=residual, extra=extra)
    torch.cuda.synchronize()
    log.append((f"conv k{conv.kernel_size} s{conv.stride} {'T' if conv.transposed else ''} {conv.in_channels}->{conv.out_channels} "
                f"ts{x.tensor_stride}->{out.tensor_stride} reps{x.replicas} rows{x.coordinate_manager.rows(out.tensor_stride)}",
                valid(out.F, out.coordinate_manager, out.tensor_stride, out.replicas)))
    return out


mn.conv_bn_act = cba
res = {}
for mode in ("twin", "free"):
    log.clear()
    pipe = DiffCompletion(denoising_steps=steps, cond_weight=6.0, device=dev)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
    pipe.read_free, pipe.hint_lag = mode == "free", mode == "twin"
    pipe.new_scheduler()
    out = pipe.completion_loop(scan, pipe.points_to_tensor(x0, role="x_t"), pipe.points_to_tensor(scan, role="cond"),
                               pipe.points_to_tensor(torch.zeros_like(scan), role="uncond"), noises=zs)
    res[mode] = (out, list(log))
a, b = res["twin"], res["free"]
print("final points differ:", int((a[0] != b[0]).any(axis=1).sum()), "of", a[0].shape[0], "ops logged", len(a[1]), len(b[1]))
for i, ((na, fa), (nb, fb)) in enumerate(zip(a[1], b[1])):
    if na != nb or fa.shape != fb.shape:
        print(i, "DIFFERENT OP / SHAPE:", na, tuple(fa.shape), "|", nb, tuple(fb.shape))
        break
    if not torch.equal(fa, fb):
        d = (fa - fb).abs()
        print(i, "first differing op:", na, "rows differing", int((d.amax(1) > 0).sum()), "of", fa.shape[0], "max", float(d.max()))
        rows = torch.nonzero(d.amax(1) > 0).flatten()[:8].tolist()
        print("   rows", rows)
        break
else:
    print("every logged operator equal")
