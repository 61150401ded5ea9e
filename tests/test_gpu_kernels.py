"""GPU parity tests (pytest -m gpu): every HIP kernel, called through the C ABI, against the CPU
oracle on the same seeded inputs and against the committed golden vectors.
Integer / index outputs are compared BIT-EXACT; fp32 feature outputs within rtol 1e-4 / atol 1e-4
(sum order differs: the oracle adds offsets in ascending k with MKL dot products, the device
accumulates an fma chain per offset).
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, noisy_scan_points, random_cloud, record_parity
from oracle import me_cpu as me

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-4


def dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def check_maps(coords_np, device):
    """voxelize + 4 stride levels + kernel maps on device vs oracle, all bit-exact."""
    from lidiff_amd import ops
    st = status(device)
    uniq, inv, first, table = ops.vox_unique(dev_i32(coords_np, device), st)
    o_uniq, o_inv, o_first = me.voxelize(coords_np)
    assert np.array_equal(uniq.cpu().numpy(), o_uniq)
    assert np.array_equal(inv.cpu().numpy(), o_inv)
    assert np.array_equal(first.cpu().numpy(), o_first)
    nbr3 = ops.kernel_map(uniq, table, 3, 1)
    assert np.array_equal(nbr3.cpu().numpy(), me.kernel_map(o_uniq, o_uniq, 3, 1))
    assert torch.equal(ops.kernel_map(uniq, table, 3, 1, self_map=True), nbr3)      # 13 lookups + mirror entries: same table
    cur, cur_t, o_cur, ts = uniq, table, o_uniq, 1
    for _ in range(4):
        coarse, parent, ctable = ops.map_stride(cur, ts * 2, st)
        o_coarse, o_parent = me.stride_map(o_cur, ts * 2)
        assert np.array_equal(coarse.cpu().numpy(), o_coarse)
        assert np.array_equal(parent.cpu().numpy(), o_parent)
        down = ops.kernel_map(coarse, cur_t, 2, ts)
        o_down = me.kernel_map(o_cur, o_coarse, 2, ts)
        assert np.array_equal(down.cpu().numpy(), o_down)
        assert torch.equal(ops.kernel_map_down(cur, parent, ts, coarse.shape[0]), down)   # from the parent array: same table
        up = ops.kernel_map_up(cur, parent, ts)
        assert np.array_equal(up.cpu().numpy(), me.transpose_kernel_map(o_down, o_cur.shape[0]))
        k3 = ops.kernel_map(coarse, ctable, 3, ts * 2)
        assert np.array_equal(k3.cpu().numpy(), me.kernel_map(o_coarse, o_coarse, 3, ts * 2))
        assert torch.equal(ops.kernel_map(coarse, ctable, 3, ts * 2, self_map=True), k3)
        cur, cur_t, o_cur, ts = coarse, ctable, o_coarse, ts * 2
    assert int(st.item()) == 0
    return nbr3


@pytest.mark.parametrize("n,extent,batch,seed", [(1, 3, 1, 0), (77, 2, 1, 1), (5000, 12, 2, 2), (40000, 40, 3, 3)])
def test_voxel_hash_maps_random(device, n, extent, batch, seed):
    check_maps(random_cloud(n, extent, seed, batch=batch), device)


def test_voxel_hash_degenerate(device):
    from lidiff_amd import ops
    check_maps(np.zeros((180, 4), np.int32), device)                      # x_uncond: one voxel
    c = random_cloud(3000, 30, 5)
    c[:, 1:] -= 500                                                        # all-negative coordinates
    check_maps(c, device)
    two = np.concatenate([random_cloud(500, 4, 6), random_cloud(500, 4, 6)])   # identical clouds, 2 batches
    two[500:, 0] = 1
    check_maps(two, device)
    st = status(device)
    uniq, inv, first, _ = ops.vox_unique(torch.zeros((0, 4), dtype=torch.int32, device=device), st)
    assert uniq.shape[0] == 0 and inv.shape[0] == 0
    bad = np.array([[0, 40000, 0, 0], [0, 1, 2, 3]], np.int32)            # outside the 16-bit key range
    ops.vox_unique(dev_i32(bad, device), st)
    assert int(st.item()) & ops.STATUS_KEY_RANGE


def test_golden_coords(device):
    from lidiff_amd import ops
    g = np.load(os.path.join(GOLDEN, "coords_small.npz"))
    st = status(device)
    uniq, inv, first, table = ops.vox_unique(dev_i32(g["coords"], device), st)
    assert np.array_equal(uniq.cpu().numpy(), g["uniq"])
    assert np.array_equal(inv.cpu().numpy(), g["inverse"])
    assert np.array_equal(first.cpu().numpy(), g["first_idx"])
    nbr = ops.kernel_map(uniq, table, 3, 1)
    assert np.array_equal(nbr.cpu().numpy(), g["nbr3_l0"])
    pin, pout, ptr = ops.rulebook_compact(nbr)
    assert np.array_equal(ptr.cpu().numpy(), g["rb_ptr"])
    assert np.array_equal(pin.cpu().numpy(), g["rb_in"]) and np.array_equal(pout.cpu().numpy(), g["rb_out"])
    cur, cur_t, ts = uniq, table, 1
    for lvl in range(1, 5):
        coarse, parent, ctable = ops.map_stride(cur, ts * 2, st)
        assert np.array_equal(coarse.cpu().numpy(), g[f"coarse{lvl}"])
        assert np.array_equal(parent.cpu().numpy(), g[f"parent{lvl}"])
        assert np.array_equal(ops.kernel_map(coarse, cur_t, 2, ts).cpu().numpy(), g[f"nbr_down{lvl}"])
        cur, cur_t, ts = coarse, ctable, ts * 2


def test_floor_and_mean(device, fps_scan):
    from lidiff_amd import ops
    pts = noisy_scan_points(fps_scan, 0.05, 0, n_rep=3)
    feats = torch.from_numpy(pts)
    cf = torch.cat([torch.zeros(pts.shape[0], 1), torch.round(feats / 0.05)], 1) - 0.25   # non-integral on purpose
    ci = ops.coords_floor(cf.to(device))
    assert np.array_equal(ci.cpu().numpy(), me.quantize_floor(cf.numpy()))
    uniq, inv, _, _ = ops.vox_unique(ci, status(device))
    out, counts = ops.vox_mean(feats.to(device), inv, uniq.shape[0])
    want = me.voxel_mean(feats, inv.cpu().numpy(), uniq.shape[0])
    assert torch.allclose(out.cpu(), want, rtol=1e-5, atol=1e-5)
    assert np.array_equal(counts.cpu().numpy(), np.bincount(inv.cpu().numpy(), minlength=uniq.shape[0]))


def test_voxel_mean_is_deterministic_and_correctly_rounded(device, fps_scan):
    """lidiff_vox_mean sums a voxel's members in 64-bit fixed point with integer atomics: the result does not depend on the
    order the atomics land in (bit-identical across runs, also with other kernels in flight), it is the EXACT sum rounded to
    fp32 once -- hence bit for bit the sequential fp32 sum (ME's CPU order, oracle/me_cpu.py:voxel_mean) wherever a voxel
    holds one or two points -- and within one rounding of the float64 mean everywhere else (heavy duplicates, the
    all-in-one-voxel x_uncond field of pipeline:89, large and tiny magnitudes in one call)."""
    from lidiff_amd import ops
    pts = noisy_scan_points(fps_scan, 0.05, 0, n_rep=10)                      # 180 000 points, ~1.1 per voxel
    feats = torch.from_numpy(pts)
    ci = torch.cat([torch.zeros(pts.shape[0], 1), torch.round(feats / 0.05)], 1).to(torch.int32).to(device)
    uniq, inv, _, _ = ops.vox_unique(ci, status(device))
    m = uniq.shape[0]
    fd = feats.to(device)
    out, counts = ops.vox_mean(fd, inv, m)
    side = torch.cuda.Stream(device=device)
    with torch.cuda.stream(side):                                             # noise on the chip while the second run sums
        junk = torch.randn(4096, 4096, device=device)
        for _ in range(3):
            junk = junk @ junk * 1e-3
    out2, _ = ops.vox_mean(fd, inv, m)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    inv_np = inv.cpu().numpy()
    cnt = np.bincount(inv_np, minlength=m)
    want = me.voxel_mean(feats, inv_np, m)                                    # sequential fp32 sums in point order
    few = torch.from_numpy(cnt <= 2)
    assert few.float().mean() > 0.8 and torch.equal(out.cpu()[few], want[few])
    exact = torch.zeros(m, 3, dtype=torch.float64).index_add_(0, torch.from_numpy(inv_np), feats.double())
    exact = (exact.float() / torch.from_numpy(cnt).float()[:, None])          # exact sum -> fp32 once -> fp32 division
    assert torch.equal(out.cpu(), exact)
    # every point in ONE voxel (x_uncond), mixed magnitudes, three runs
    n = 70001
    g = torch.Generator().manual_seed(4)
    vals = torch.randn(n, 3, generator=g) * torch.tensor([50.0, 1e-3, 1.0])
    one = torch.zeros(n, dtype=torch.int64, device=device)
    runs = [ops.vox_mean(vals.to(device), one, 1)[0].cpu() for _ in range(3)]
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    ref = vals.double().mean(0)
    assert torch.allclose(runs[0][0].double(), ref, rtol=1e-6, atol=1e-7), (runs[0], ref)
    zeros = ops.vox_mean(torch.zeros(n, 3, device=device), one, 1)[0]
    assert torch.equal(zeros.cpu(), torch.zeros(1, 3))


def test_voxel_mean_confines_a_non_finite_feature_to_its_voxel(device, fps_scan):
    """An Inf / NaN feature poisons the voxel (and channel) it belongs to, as the reference's fp32 sum does, and nothing else
    (the common fixed-point scale made the whole batch NaN): every other voxel keeps the bits of the clean run; the
    poisoned channels read what the sequential fp32 sum gives (NaN, +-Inf, Inf - Inf = NaN)."""
    from lidiff_amd import ops
    pts = noisy_scan_points(fps_scan, 0.05, 0, n_rep=2)
    feats = torch.from_numpy(pts).clone()
    ci = torch.cat([torch.zeros(pts.shape[0], 1), torch.round(feats / 0.2)], 1).to(torch.int32).to(device)   # several points per voxel
    uniq, inv, _, _ = ops.vox_unique(ci, status(device))
    m = uniq.shape[0]
    clean, _ = ops.vox_mean(feats.to(device), inv, m)
    inv_np = inv.cpu().numpy()
    order = np.argsort(np.bincount(inv_np, minlength=m))[::-1]
    va, vb, vc = (int(v) for v in order[:3])                                   # three crowded voxels
    pa, pb, pc = (np.flatnonzero(inv_np == v) for v in (va, vb, vc))
    assert min(len(pa), len(pb), len(pc)) >= 2
    feats[pa[0], 0] = float("nan")
    feats[pb[0], 1] = float("inf")
    feats[pc[0], 2] = float("inf")
    feats[pc[1], 2] = float("-inf")
    out, counts = ops.vox_mean(feats.to(device), inv, m)
    out, clean = out.cpu(), clean.cpu()
    untouched = torch.ones(m, dtype=torch.bool)
    untouched[[va, vb, vc]] = False
    assert torch.equal(out[untouched], clean[untouched])
    assert torch.isnan(out[va, 0]) and torch.equal(out[va, 1:], clean[va, 1:])
    assert out[vb, 1] == float("inf") and torch.equal(out[vb, [0, 2]], clean[vb, [0, 2]])
    assert torch.isnan(out[vc, 2]) and torch.equal(out[vc, :2], clean[vc, :2])
    assert np.array_equal(counts.cpu().numpy(), np.bincount(inv_np, minlength=m))


def conv_case(device, coords_np, cin, cout, kind, seed, epilogue=False, split=0):
    """kind: 'k3' | 'down' | 'up' | 'k1'."""
    from lidiff_amd import ops
    uniq, _, _ = me.voxelize(coords_np)
    coarse, parent = me.stride_map(uniq, 2)
    if kind == "k3":
        nbr, m_in, m_out, K = me.kernel_map(uniq, uniq, 3, 1), uniq.shape[0], uniq.shape[0], 27
    elif kind == "down":
        nbr, m_in, m_out, K = me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0], coarse.shape[0], 8
    elif kind == "up":
        nbr = me.transpose_kernel_map(me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0])
        m_in, m_out, K = coarse.shape[0], uniq.shape[0], 8
    else:
        nbr, m_in, m_out, K = None, uniq.shape[0], uniq.shape[0], 1
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) / np.sqrt(cin * max(1, K // 3))
    want = me.conv_forward(x.double(), w.double() if K > 1 else w[0].double(), nbr)
    scale = shift = res = None
    relu = False
    if epilogue:
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = torch.randn(m_out, cout, generator=g)
        relu = True
        want = torch.relu(want * scale.double() + shift.double() + res.double())
    d = lambda t: None if t is None else t.to(device)
    nbr_d = None if nbr is None else dev_i32(nbr, device)
    for hint in (False, True):          # dense-map kernel and the low-density (packed-stage) kernel: same results
        if split:
            got = ops.spconv_fwd(d(x[:, :split].contiguous()), d(w), nbr_d, m_out, in_b=d(x[:, split:].contiguous()),
                                 scale=d(scale), shift=d(shift), residual=d(res), relu=relu, sparse_map=hint)
        else:
            got = ops.spconv_fwd(d(x), d(w), nbr_d, m_out, scale=d(scale), shift=d(shift), residual=d(res), relu=relu,
                                 sparse_map=hint)
        torch.cuda.synchronize()
        err = (got.cpu().double() - want).abs().max().item()
        assert torch.allclose(got.cpu().double(), want, rtol=RTOL, atol=ATOL), f"{kind} {cin}->{cout} hint={hint}: max err {err}"


@pytest.mark.parametrize("cin,cout", [(3, 32), (32, 32), (32, 64), (64, 128), (96, 96), (128, 256), (384, 256)])
def test_spconv_k3_channels(device, cin, cout):
    conv_case(device, random_cloud(3000, 6, cin + cout, batch=2), cin, cout, "k3", seed=cin)


@pytest.mark.parametrize("kind,cin,cout", [("down", 32, 32), ("down", 128, 128), ("up", 256, 256), ("up", 128, 96),
                                           ("k1", 32, 64), ("k1", 384, 256), ("k1", 128, 96)])
def test_spconv_other_kinds(device, kind, cin, cout):
    conv_case(device, random_cloud(2500, 7, 17), cin, cout, kind, seed=3)


def test_spconv_epilogue_and_split_input(device):
    c = random_cloud(2000, 5, 23)
    conv_case(device, c, 64, 64, "k3", seed=1, epilogue=True)
    conv_case(device, c, 192, 128, "k3", seed=2, epilogue=True, split=128)      # fused ME.cat (128 | 64)
    conv_case(device, c, 128, 96, "k1", seed=3, epilogue=True, split=96)         # downsample conv on cat
    conv_case(device, c, 3, 32, "k3", seed=4, epilogue=True)                      # scalar (non-float4) path


@pytest.mark.parametrize("cin,cout", [(32, 32), (96, 96), (128, 96), (64, 128), (128, 64)])
def test_spconv_low_density_maps(device, cin, cout):
    """Isolated voxels (about one neighbour per voxel, like the stride-1/2 levels of a noisy scan): the tiles of the
    32-channel kernel take the packed-stage path (several offsets per 128-row stage)."""
    conv_case(device, random_cloud(6000, 40, cin * 7 + cout, batch=2, dup=0.05), cin, cout, "k3", seed=cin + 1)
    conv_case(device, random_cloud(3000, 20, 5, dup=0.0), cin, cout, "down", seed=2)
    conv_case(device, random_cloud(3000, 20, 6, dup=0.0), cin, cout, "up", seed=3)


def test_spconv_row_order(device):
    """Morton-ordered tiles (row_order + permuted table columns): identical results, epilogue included."""
    from lidiff_amd import ops
    coords = random_cloud(5000, 9, 43, batch=2)
    uniq, _, _ = me.voxelize(coords)
    nbr_np = me.kernel_map(uniq, uniq, 3, 1)
    m = uniq.shape[0]
    order = ops.tile_order(dev_i32(uniq, device), 1)
    assert sorted(order.cpu().tolist()) == list(range(m))
    key = lambda c: [int(v) for v in c]
    # Morton order: consecutive rows are close in space (mean L1 step far below a random permutation's)
    cu = torch.from_numpy(uniq[:, 1:]).float().to(device)
    step_sorted = (cu[order.long()][1:] - cu[order.long()][:-1]).abs().sum(1).mean().item()
    step_plain = (cu[1:] - cu[:-1]).abs().sum(1).mean().item()
    assert step_sorted < 0.5 * step_plain
    nbr = dev_i32(nbr_np, device)
    nbr_o = nbr.index_select(1, order.long()).contiguous()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2 * m, 64, generator=g).to(device)
    w = (torch.randn(27, 64, 128, generator=g) * 0.1).to(device)
    res = torch.randn(2 * m, 128, generator=g).to(device)
    sc, sh = (torch.rand(128, generator=g) + 0.5).to(device), torch.randn(128, generator=g).to(device)
    plain = ops.spconv_fwd(x, w, nbr, m, scale=sc, shift=sh, residual=res, relu=True, replicas=2)
    ordered = ops.spconv_fwd(x, w, nbr_o, m, scale=sc, shift=sh, residual=res, relu=True, replicas=2, row_order=order)
    assert torch.allclose(plain, ordered, rtol=1e-5, atol=1e-5)
    w1 = (torch.randn(1, 64, 32, generator=g) * 0.1).to(device)                   # identity map through the order
    assert torch.allclose(ops.spconv_fwd(x[:m], w1, None, m), ops.spconv_fwd(x[:m], w1, None, m, row_order=order), atol=1e-6)


def test_spconv_replicas(device):
    """R stacked feature matrices over one kernel map (the CFG pair): one launch == R launches."""
    from lidiff_amd import ops
    coords = random_cloud(3000, 6, 41, batch=2)
    uniq, _, _ = me.voxelize(coords)
    nbr = dev_i32(me.kernel_map(uniq, uniq, 3, 1), device)
    m = uniq.shape[0]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2 * m, 64, generator=g).to(device)
    w = (torch.randn(27, 64, 128, generator=g) * 0.1).to(device)
    res = torch.randn(2 * m, 128, generator=g).to(device)
    sc, sh = (torch.rand(128, generator=g) + 0.5).to(device), torch.randn(128, generator=g).to(device)
    both = ops.spconv_fwd(x, w, nbr, m, scale=sc, shift=sh, residual=res, relu=True, replicas=2)
    for r in range(2):
        one = ops.spconv_fwd(x[r * m:(r + 1) * m], w, nbr, m, scale=sc, shift=sh, residual=res[r * m:(r + 1) * m], relu=True)
        assert torch.equal(both[r * m:(r + 1) * m], one)


def test_spconv_degenerate_shapes(device):
    conv_case(device, np.zeros((5, 4), np.int32), 32, 32, "k3", seed=0)          # a single voxel
    conv_case(device, random_cloud(129, 50, 1, dup=0.0), 32, 32, "k3", seed=0)  # isolated voxels, ragged tile
    conv_case(device, random_cloud(4000, 2, 2), 64, 64, "k3", seed=0)            # dense 5^3 block: all 27 neighbours


def test_golden_conv(device):
    from lidiff_amd import ops
    g = np.load(os.path.join(GOLDEN, "coords_small.npz"))
    cv = np.load(os.path.join(GOLDEN, "conv_small.npz"))
    d = lambda a: torch.from_numpy(a).to(device)
    m0, m1 = g["uniq"].shape[0], g["coarse1"].shape[0]
    y3 = ops.spconv_fwd(d(cv["x0"]), d(cv["w3"]), dev_i32(g["nbr3_l0"], device), m0)
    yd = ops.spconv_fwd(d(cv["x0"]), d(cv["w2"]), dev_i32(g["nbr_down1"], device), m1)
    yu = ops.spconv_fwd(yd, d(cv["wt"]), dev_i32(cv["nbr_up1"], device), m0)
    y1 = ops.spconv_fwd(d(cv["x0"]), d(cv["w1"]), None, m0)
    for got, name in ((y3, "y3"), (yd, "yd"), (yu, "yu"), (y1, "y1")):
        assert torch.allclose(got.cpu(), torch.from_numpy(cv[name]), rtol=RTOL, atol=ATOL), name


def test_spconv_linearity_full_size(device, fps_scan):
    """Size-independent property at BASELINE's full size (180k points): conv(a*x + y) ==
    a*conv(x) + conv(y), and a one-hot centre kernel is the identity."""
    from lidiff_amd import ops
    pts = noisy_scan_points(fps_scan, 0.2, 1)
    cf = torch.cat([torch.zeros(pts.shape[0], 1), torch.round(torch.from_numpy(pts) / 0.05)], 1)
    ci = ops.coords_floor(cf.to(device))
    uniq, _, _, table = ops.vox_unique(ci, status(device))
    m = uniq.shape[0]
    assert 150000 < m <= 180000
    nbr = ops.kernel_map(uniq, table, 3, 1)
    # symmetry of the kernel map: nbr[k][o] = i  <=>  nbr[26-k][i] = o
    k = 5
    o = torch.nonzero(nbr[k] >= 0).squeeze(1)
    assert torch.equal(nbr[26 - k][nbr[k][o].long()].long(), o)
    assert torch.equal(nbr[13], torch.arange(m, dtype=torch.int32, device=device))
    g = torch.Generator(device="cpu").manual_seed(0)
    x, y = torch.randn(m, 32, generator=g).to(device), torch.randn(m, 32, generator=g).to(device)
    w = (torch.randn(27, 32, 32, generator=g) * 0.1).to(device)
    lhs = ops.spconv_fwd(2.5 * x + y, w, nbr, m)
    rhs = 2.5 * ops.spconv_fwd(x, w, nbr, m) + ops.spconv_fwd(y, w, nbr, m)
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-4)
    w_id = torch.zeros(27, 32, 32, device=device)
    w_id[13] = torch.eye(32, device=device)
    assert torch.equal(ops.spconv_fwd(x, w_id, nbr, m), x)
    # determinism: no atomics in the conv path
    assert torch.equal(ops.spconv_fwd(x, w, nbr, m), ops.spconv_fwd(x, w, nbr, m))


@pytest.mark.parametrize("kind,ks,stride,cin,cout", [("conv", 3, 1, 32, 64), ("conv", 2, 2, 64, 64), ("tconv", 2, 2, 64, 32),
                                                    ("conv", 1, 1, 96, 32), ("conv", 3, 1, 256, 256),
                                                    ("conv", 3, 1, 192, 144), ("conv", 1, 1, 320, 256)])
def test_spconv_backward_vs_oracle_autograd(device, kind, ks, stride, cin, cout):
    """Training path (models.py:180-217): dX (the same HIP kernel over the swapped map with W^T) and dW
    (lidiff_spconv_bwd_w over the map's rulebook: every tile instantiation, partial ci / co tiles, the identity map)
    of the ME-shim convolutions against torch autograd through the oracle."""
    import lidiff_amd.MinkowskiEngine as ME
    coords = random_cloud(1500, 5, 31, batch=2)
    g = torch.Generator().manual_seed(7)
    uniq, _, _ = me.voxelize(coords)
    field = ME.TensorField(features=torch.randn(coords.shape[0], 3, generator=g).to(device),
                           coordinates=torch.from_numpy(coords).float().to(device), device=device)
    x0 = field.sparse()
    mgr = x0.coordinate_manager
    ts_in = 1
    if kind == "tconv":                                   # needs the coarse map and an input living on it
        mgr.stride(1, 2)
        ts_in = 2
    m_in = mgr.maps[ts_in].coords.shape[0]
    xf = torch.randn(m_in, cin, generator=g)
    x = ME.SparseTensor(xf.to(device).requires_grad_(True), tensor_stride=ts_in, coordinate_manager=mgr)
    mod = (ME.MinkowskiConvolutionTranspose if kind == "tconv" else ME.MinkowskiConvolution)(
        cin, cout, kernel_size=ks, stride=stride, dimension=3).to(device)
    y = mod(x)
    r = torch.randn(y.F.shape, generator=g)
    (y.F * r.to(device)).sum().backward()
    # oracle: same maps on the CPU, autograd through torch ops
    coarse, _ = me.stride_map(uniq, 2)
    if kind == "tconv":
        nbr = me.transpose_kernel_map(me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0])
    elif ks == 1:
        nbr = None
    elif stride == 2:
        nbr = me.kernel_map(uniq, coarse, 2, 1)
    else:
        nbr = me.kernel_map(uniq, uniq, 3, 1)
    xo = xf.double().requires_grad_(True)
    wo = mod.kernel.detach().cpu().double().requires_grad_(True)
    yo = me.conv_forward(xo, wo, nbr)
    assert torch.allclose(y.F.detach().cpu().double(), yo.detach(), rtol=RTOL, atol=ATOL)
    (yo * r.double()).sum().backward()
    assert torch.allclose(x.F.grad.cpu().double(), xo.grad, rtol=1e-4, atol=1e-4), "dX"
    assert torch.allclose(mod.kernel.grad.cpu().double(), wo.grad, rtol=1e-4, atol=1e-3), "dW"
    # dW is bit-reproducible (pair slices summed in slice order through a workspace), and the atomic form agrees with it
    from lidiff_amd import ops
    g1 = mod.kernel.grad.clone()
    mod.kernel.grad = None
    x.F.grad = None
    (mod(x).F * r.to(device)).sum().backward()
    assert torch.equal(mod.kernel.grad, g1), "dW not deterministic"
    ops.DETERMINISTIC_DW = False
    try:
        mod.kernel.grad = None
        (mod(x).F * r.to(device)).sum().backward()
        assert torch.allclose(mod.kernel.grad, g1, rtol=1e-4, atol=1e-4), "atomic dW"
    finally:
        ops.DETERMINISTIC_DW = True


def test_stem_weight_gradient_pads_the_input_columns(device):
    """dW of the 3-channel stem convolution (minkunet.py:155: in_channels = 3): ops.spconv_bwd_w zero-pads the input columns
    to 4 and runs the MFMA kernel; against torch autograd through the oracle's convolution (float64)."""
    from lidiff_amd import ops
    coords = random_cloud(3000, 7, 31, batch=2)
    uniq, _, _ = me.voxelize(coords)
    nbr = me.kernel_map(uniq, uniq, 3, 1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(uniq.shape[0], 3, generator=g)
    w = (torch.randn(27, 3, 32, generator=g) / 3.0).double().requires_grad_(True)
    up = torch.randn(uniq.shape[0], 32, generator=g)
    (me.conv_forward(x.double(), w, nbr) * up.double()).sum().backward()
    got = ops.spconv_bwd_w(x.to(device), up.to(device), dev_i32(nbr, device), 27)
    assert got.shape == (27, 3, 32)
    assert torch.allclose(got.cpu().double(), w.grad, rtol=RTOL, atol=ATOL), (got.cpu().double() - w.grad).abs().max().item()


def test_gather_scatter_rows(device):
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(0)
    for c in (3, 96):
        src = torch.randn(500, c, generator=g)
        idx = torch.randint(0, 500, (2000,), generator=g)
        assert torch.equal(ops.gather_rows(src.to(device), idx.to(device)).cpu(), src[idx])
        vals = torch.randn(2000, c, generator=g)
        want = torch.zeros(500, c).index_add_(0, idx, vals)
        assert torch.allclose(ops.scatter_add_rows(vals.to(device), idx.to(device), 500).cpu(), want, atol=1e-4)


def test_farthest_point_sample(device, fps_scan):
    """preprocess_scan (pipeline:92-105): the HIP FPS must pick the same indices as the float64 torch loop
    (open3d semantics: start at 0, first maximum), duplicates and ties included."""
    from lidiff_amd import ops
    from lidiff_amd.pipeline import farthest_point_sample
    g = torch.Generator().manual_seed(4)
    pts = torch.randn(5000, 3, generator=g, dtype=torch.float64) * torch.tensor([20.0, 20.0, 1.0], dtype=torch.float64)
    pts[100:200] = pts[0:100]                                   # exact duplicates: zero distances, index ties
    want = farthest_point_sample(pts, 400)
    got = ops.farthest_point_sample(pts.to(device), 400).cpu()
    assert torch.equal(got, want)
    # lattice points: many exactly equal distances -> the first maximum must win
    grid = torch.stack(torch.meshgrid(torch.arange(12.0), torch.arange(12.0), torch.arange(3.0), indexing="ij"), -1)
    grid = grid.reshape(-1, 3).double()
    assert torch.equal(ops.farthest_point_sample(grid.to(device), 100).cpu(), farthest_point_sample(grid, 100))
    # the bundled scan's 18 000 samples, re-sampled: still a subset in the same order on both sides
    sub = torch.from_numpy(fps_scan.astype(np.float64))
    assert torch.equal(ops.farthest_point_sample(sub.to(device), 2000).cpu(), farthest_point_sample(sub, 2000))
    assert torch.equal(ops.farthest_point_sample(sub.to(device), 18000).cpu(), torch.arange(18000))


def test_gather_mul_rows(device):
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(2)
    for n, mp, c in ((1, 1, 32), (3001, 700, 96), (20000, 5824, 256)):
        x, table = torch.randn(n, c, generator=g), torch.randn(mp, c, generator=g)
        idx = torch.randint(0, mp, (n,), generator=g)
        got = ops.gather_mul_rows(x.to(device), table.to(device), idx.to(device))
        assert torch.equal(got.cpu(), x * table[idx])                  # one multiply per element: bit-exact
    buf = torch.zeros(10, 32, device=device)
    ops.gather_mul_rows(torch.ones(4, 32, device=device), torch.full((2, 32), 3.0, device=device),
                        torch.tensor([0, 1, 1, 0], device=device), out=buf[3:7])
    assert float(buf.sum()) == 4 * 32 * 3.0 and float(buf[:3].abs().sum() + buf[7:].abs().sum()) == 0.0


def test_gather_bias_leaky(device):
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(2)
    src, bias = torch.randn(300, 96, generator=g), torch.randn(1, 96, generator=g)
    idx = torch.randint(0, 300, (5000,), generator=g)
    want = torch.nn.functional.leaky_relu(src[idx] + bias, 0.1)
    buf = torch.zeros(2 * 5000, 96, device=device)
    got = ops.gather_bias_leaky(src.to(device), idx.to(device), bias.to(device), 0.1, out=buf[5000:])
    assert torch.equal(got.cpu(), want) and torch.equal(buf[5000:].cpu(), want) and float(buf[:5000].abs().sum()) == 0.0


def test_nn_match(device):
    from lidiff_amd import ops
    full = random_cloud(5000, 60, 3, batch=2)
    part = me.voxelize(me.floor_to_stride(random_cloud(1500, 60, 4, batch=2), 16))[0]
    got = ops.nn_match(dev_i32(full, device), dev_i32(part, device)).cpu().numpy()
    assert np.array_equal(got, me.argmin_match(full, part))
    # more part rows than one LDS tile -> the part rows are split over blockIdx.y and merged by atomic min;
    # duplicated part rows put exact ties into different splits (lowest row must still win)
    part_dup = np.tile(me.floor_to_stride(random_cloud(1700, 40, 9, batch=2), 16), (3, 1))
    for n_full in (300, 5000, 70000):
        fc = random_cloud(n_full, 60, 10 + n_full, batch=2)
        got = ops.nn_match(dev_i32(fc, device), dev_i32(part_dup, device)).cpu().numpy()
        assert np.array_equal(got, me.argmin_match(fc, part_dup)), n_full
    one = np.zeros((1, 4), np.int32)                                   # x_uncond: a single part voxel
    assert np.all(ops.nn_match(dev_i32(full, device), dev_i32(one, device)).cpu().numpy() == 0)
    f = np.array([[0, 0, 0, 0]], np.int32)                             # exact tie -> lowest index
    p = np.array([[0, 2, 0, 0], [0, -2, 0, 0], [0, 0, 2, 0]], np.int32)
    assert ops.nn_match(dev_i32(f, device), dev_i32(p, device)).item() == 0
    # by batch element first (training batches, lidiff_nn_match d_gate): the same winners -- part rows grouped by batch (a
    # voxelised batch: the usual case, the restricted pass is final), interleaved batches (the check sees that the rows are not
    # grouped: the unrestricted pass runs), rows farther from every part row of their own element than the batch term (not
    # conclusive: unrestricted pass; the winner may sit in ANOTHER element, exactly as pykeops would find it), a batch element
    # without part rows, three elements
    def grouped(rows):
        return rows[np.argsort(rows[:, 0], kind="stable")]
    cases = [(grouped(random_cloud(70000, 60, 31, batch=2)), grouped(part)),
             (grouped(random_cloud(70000, 60, 32, batch=3)), grouped(me.voxelize(me.floor_to_stride(random_cloud(4000, 60, 33, batch=3), 8))[0])),
             (random_cloud(9000, 60, 34, batch=2), part_dup),                                   # not grouped at all
             (grouped(random_cloud(9000, 60, 35, batch=2)), grouped(part)[grouped(part)[:, 0] == 0])]    # element 1 has no part row
    far = grouped(random_cloud(3000, 60, 36, batch=2))
    far[::7, 1] += 4000                                                                          # rows far away from everything
    cases.append((far, grouped(part)))
    for fc, pc in cases:
        got = ops.nn_match(dev_i32(fc, device), dev_i32(pc, device), by_batch=True).cpu().numpy()
        assert np.array_equal(got, me.argmin_match(fc, pc)), (fc.shape, pc.shape)
        assert np.array_equal(got, ops.nn_match(dev_i32(fc, device), dev_i32(pc, device)).cpu().numpy())
    # row count on the device (the matches queued inside a pyramid chain, lidiff_nn_match_dev): a buffer sized for a BOUND whose
    # rows beyond the count hold garbage (larger coordinates than any valid row: they must not enter max_coord) -- same winners
    for n_full, n_bound in ((300, 5000), (5000, 5000), (70000, 180000), (1, 64)):
        fc = random_cloud(n_full, 60, 20 + n_full, batch=2)
        buf = np.full((n_bound, 4), 30000, np.int32)
        buf[:n_full] = fc
        cnt = torch.tensor([n_full], dtype=torch.int32, device=device)
        for pc in (part, part_dup, one):
            got = ops.nn_match_dev(dev_i32(buf, device), cnt, dev_i32(pc, device))[:n_full].cpu().numpy()
            assert np.array_equal(got, me.argmin_match(fc, pc)), (n_full, n_bound, pc.shape)


def test_step_boundary_kernels_edge_cases(device):
    """lidiff_points_to_field / lidiff_cfg_dpm_step (step.hip) and lidiff_nn_match_dev on the inputs the happy path never sees: no
    points, ties of the rounding (x * 20 exactly half-way: round half to EVEN, as torch.round does), negative coordinates, a
    second-order update whose previous prediction is given, no noise term, and a match over a map with zero valid rows."""
    from lidiff_amd import ops
    from lidiff_amd.schedulers import DPMSolverMultistepScheduler
    f, c = ops.points_to_field(torch.zeros((1, 0, 3), dtype=torch.float64, device=device), 0.05)
    assert f.shape == (0, 3) and c.shape == (0, 4)
    pts = torch.tensor([[[0.025, 0.075, -0.025], [-0.075, 0.125, 1e-9], [81.9175, -81.92, 0.0499999]]], dtype=torch.float64, device=device)
    f, c = ops.points_to_field(pts, 0.05)
    want = torch.round(pts[0].float() * 20.0).to(torch.int32)
    assert torch.equal(c[:, 1:], want) and torch.equal(c[:, 0], torch.zeros(3, dtype=torch.int32, device=device))
    assert c[0].tolist() == [0, 0, 2, 0] and c[1].tolist()[:3] == [0, -2, 2]          # half-way cases go to the even neighbour
    s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=3.5e-5, beta_end=0.007, beta_schedule="linear",
                                    algorithm_type="sde-dpmsolver++", solver_order=2)
    s.set_timesteps(50)
    s.to(device)
    x_init = torch.zeros((1, 0, 3), dtype=torch.float64, device=device)
    e = torch.zeros((1, 0, 3), device=device)
    x0, f, c = ops.cfg_dpm_step(e, e, 6.0, torch.zeros((0, 3), device=device), x_init, s.step_plan(999), None, 0.05)
    assert x0.shape == (1, 0, 3) and f.shape == (0, 3) and c.shape == (0, 4)
    # one point, no noise term, first then second order: against the scheduler's own torch arithmetic
    g = torch.Generator().manual_seed(1)
    x_init = torch.randn(1, 7, 3, generator=g, dtype=torch.float64).to(device) * 10
    x_t = (x_init[0] + torch.randn(7, 3, generator=g, dtype=torch.float64).to(device)).float()
    s2 = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_start=3.5e-5, beta_end=0.007, beta_schedule="linear",
                                     algorithm_type="sde-dpmsolver++", solver_order=2)
    s2.set_timesteps(50)
    s2.to(device)
    s.set_timesteps(50)
    xa = xb = x_t
    for t in s.host_timesteps[:3]:
        ec, eu = (torch.randn(1, 7, 3, generator=g).to(device) for _ in range(2))
        z = torch.zeros(1, 7, 3, dtype=torch.float64, device=device)
        x0, fa, ca = ops.cfg_dpm_step(ec, eu, 6.0, xa, x_init, s.step_plan(t), z, 0.05)
        s.commit(x0)
        prev = s2.step(eu + 6.0 * (ec - eu), t, xb.reshape(1, -1, 3) - x_init, noise=z)["prev_sample"]
        fb = (x_init + prev).float()[0]
        assert torch.equal(fa, fb) and torch.equal(ca[:, 1:], torch.round(fb / 0.05).to(torch.int32)), t
        xa, xb = fa, fb
    # a map with zero valid rows (count 0 on the device): nothing is written, nothing faults
    part = dev_i32(np.array([[0, 1, 2, 3], [0, 5, 5, 5]], np.int32), device)
    buf = dev_i32(np.full((500, 4), 7, np.int32), device)
    idx = ops.nn_match_dev(buf, torch.zeros(1, dtype=torch.int32, device=device), part)
    torch.cuda.synchronize()
    assert idx.shape == (500,)


def test_fps_reproduces_the_committed_scan_from_the_range_filtered_input(device, fps_scan):
    """preprocess_scan's FPS (pipeline:97-99) on the range-filtered bundled scan (119 035 points)
    through the HIP kernel selects exactly the committed 18 000 points, in order; spot-checked against the FPS oracle."""
    from lidiff_amd import ops
    from oracle.fps_cpu import farthest_point_sample as fps_oracle
    pts = np.load(os.path.join(GOLDEN, "scan_000123_range_filtered.npy")).astype(np.float64)
    import time
    pts_d = torch.from_numpy(pts).to(device)
    took = {}
    for coop in (True, False):            # the persistent cooperative kernel and the launch-per-selection kernel
        ops.FPS_COOPERATIVE = coop
        try:
            ops.farthest_point_sample(pts_d, 64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sel = ops.farthest_point_sample(pts_d, 18000).cpu().numpy()
            took[coop] = time.perf_counter() - t0
        finally:
            ops.FPS_COOPERATIVE = True
        assert np.array_equal(pts[sel].astype(np.float32), fps_scan), f"cooperative={coop}"
        assert np.array_equal(sel[:300], fps_oracle(pts, 300))
    print(f"FPS 119 035 -> 18 000: cooperative launch {1e3 * took[True]:.1f} ms, one launch per selection {1e3 * took[False]:.1f} ms")


def test_sparse_quantize_vs_oracle(device):
    """ME.utils.sparse_quantize (map_from_scans.py:91: float64 world coordinates / 0.1 m; SemanticKITTITemporalAggr.py:87):
    unique voxels in first-occurrence order, kept-row index, inverse map; numpy in -> numpy out, torch in -> torch out;
    floor in the SOURCE dtype (a float32 detour changes voxels at map scale)."""
    import lidiff_amd.MinkowskiEngine as ME
    rng = np.random.default_rng(0)
    pts = rng.uniform(-400.0, 400.0, size=(50000, 3))                       # metres, float64, map scale
    pts[1000:2000] = pts[:1000] + 1e-9                                      # duplicates inside a voxel
    q = 0.1
    k = np.arange(3000)
    pts[5000:8000, 0] = (k - 1500) * q - 1e-12                              # just BELOW voxel boundaries: float32 would
    c_want = np.floor(pts / q).astype(np.int32)                             # round them up into the next voxel
    assert (np.floor((pts / q).astype(np.float32)).astype(np.int32) != c_want).any()
    uniq_o, inv_o, first_o = me.voxelize(np.concatenate([np.zeros((len(pts), 1), np.int32), c_want], 1))
    coords, index, inverse = ME.utils.sparse_quantize(pts, return_index=True, return_inverse=True, quantization_size=q)
    assert isinstance(coords, np.ndarray) and coords.dtype == np.int32
    assert np.array_equal(coords, uniq_o[:, 1:]) and np.array_equal(index, first_o) and np.array_equal(inverse, inv_o)
    # already-quantised integer coordinates, torch in -> torch out, features follow the kept rows
    ci = torch.from_numpy(c_want[:4000])
    feats = torch.arange(4000, dtype=torch.float32)[:, None]
    c2, f2, i2 = ME.utils.sparse_quantize(ci, features=feats, return_index=True)
    u2, _, first2 = me.voxelize(np.concatenate([np.zeros((4000, 1), np.int32), c_want[:4000]], 1))
    assert torch.is_tensor(c2) and np.array_equal(c2.cpu().numpy(), u2[:, 1:])
    assert np.array_equal(i2.cpu().numpy(), first2) and np.array_equal(f2.cpu().numpy()[:, 0], first2.astype(np.float32))
    only = ME.utils.sparse_quantize(pts[:100], quantization_size=q)
    assert isinstance(only, np.ndarray) and only.shape[1] == 3
    with pytest.raises(RuntimeError):
        ME.utils.sparse_quantize(np.array([[0.0, 0.0, 40000.0]]))


def test_single_read_pyramid_equals_the_map_by_map_build(device):
    """ops.build_pyramid / CoordinateManager.pyramid -- voxelise, four strided maps, the kernel_size-3 self maps and tail-map
    counts of the first two levels queued with the row counts on the device and ONE host read -- against the map-by-map
    build (one read per map): every coordinate row, parent array, hash lookup result, neighbour table and tail map
    identical, on dense / sparse / one-voxel / two-batch clouds; and the oracle's maps for one of them."""
    import lidiff_amd.MinkowskiEngine as ME
    one = np.zeros((180, 4), np.int32)
    tiny = random_cloud(3, 50, 8)
    for cloud in (random_cloud(40000, 40, 3, batch=3), random_cloud(9000, 300, 41, dup=0.0), one, tiny,
                  random_cloud(5000, 6, 42, batch=2)):
        mgrs = []
        for pyramid in (False, True):
            mgr = ME.CoordinateManager(device)
            mgr.pyramid = pyramid
            inv, first = mgr.insert(dev_i32(cloud, device))
            ts = 1
            for _ in range(4):
                ts = mgr.stride(ts, 2)
            mgrs.append((mgr, inv, first))
        (a, inv_a, first_a), (b, inv_b, first_b) = mgrs
        assert torch.equal(inv_a, inv_b) and torch.equal(first_a, first_b)
        for ts in (1, 2, 4, 8, 16):
            assert torch.equal(a.maps[ts].coords, b.maps[ts].coords), ts
            if ts > 1:
                assert torch.equal(a.parents[ts], b.parents[ts]), ts
            assert torch.equal(a.kernel_map(ts, ts, 3), b.kernel_map(ts, ts, 3)), ts        # looks the rows up in the tables
            if ts < 16:
                assert torch.equal(a.kernel_map(ts, 2 * ts, 2), b.kernel_map(ts, 2 * ts, 2))
                assert torch.equal(a.kernel_map(2 * ts, ts, 2, True), b.kernel_map(2 * ts, ts, 2, True))
        for ts in (1, 2):
            ta, tb = a.tail_map(ts), b.tail_map(ts)
            assert ta.n == tb.n and torch.equal(ta.ptr, tb.ptr) and torch.equal(ta.off, tb.off)
            if ta.n:
                assert torch.equal(ta.nbr, tb.nbr) and torch.equal(ta.idx, tb.idx) and torch.equal(ta.pair_in, tb.pair_in)
        a.check(); b.check()
    o_uniq, o_inv, _ = me.voxelize(cloud)
    assert np.array_equal(b.maps[1].coords.cpu().numpy(), o_uniq) and np.array_equal(inv_b.cpu().numpy(), o_inv)
    o_c2, o_par = me.stride_map(o_uniq, 2)
    assert np.array_equal(b.maps[2].coords.cpu().numpy(), o_c2) and np.array_equal(b.parents[2].cpu().numpy(), o_par)
    assert np.array_equal(b.kernel_map(1, 1, 3).cpu().numpy(), me.kernel_map(o_uniq, o_uniq, 3, 1))


@pytest.mark.parametrize("cin", [3, 4, 1])
def test_spconv_thin_input_kernel_vs_oracle_and_tile_kernel(device, cin):
    """The stems' convolution (C_in <= 4 -> 32: spconv_thin_kernel, a VALU kernel over the neighbour table) against the
    float64 oracle and the tile kernel (equal up to fp32 summation order), kernel_size 3 and 1, ragged row counts, replicas,
    every epilogue combination."""
    from lidiff_amd import _lib, ops
    assert _lib.load().lidiff_spconv_fwd_kernel_id(cin, 0, 32, 27, 1, 0, 0) == 3
    assert _lib.load().lidiff_spconv_fwd_kernel_id(cin, 0, 32, 27, 1, 0, 8) == 0 and _lib.load().lidiff_spconv_fwd_kernel_id(cin, 0, 64, 27, 1, 0, 0) == 0
    g = torch.Generator().manual_seed(90 + cin)
    for cloud, reps, epi in ((random_cloud(7000, 12, 51, batch=2), 2, 3), (random_cloud(300, 3, 52), 1, 0),
                             (random_cloud(5, 40, 53), 2, 1), (random_cloud(20000, 60, 54), 1, 2)):
        uniq, _, _ = me.voxelize(cloud)
        m = uniq.shape[0]
        for k in (27, 1):
            nbr_np = me.kernel_map(uniq, uniq, 3, 1) if k == 27 else None
            nbr = None if nbr_np is None else dev_i32(nbr_np, device)
            x = torch.randn(reps * m, cin, generator=g)
            w = torch.randn(k, cin, 32, generator=g) / np.sqrt(cin * max(1, k // 3))
            sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
            res = torch.randn(reps * m, 32, generator=g)
            kw = dict(replicas=reps, scale=sc.to(device) if epi & 1 else None, shift=sh.to(device) if epi & 1 else None,
                      residual=res.to(device) if epi & 2 else None, relu=bool(epi & 2))
            got = ops.spconv_fwd(x.to(device), w.to(device), nbr, m, **kw)
            assert torch.equal(got, ops.spconv_fwd(x.to(device), w.to(device), nbr, m, **kw))
            ref = ops.spconv_fwd(x.to(device), w.to(device), nbr, m, kernel="tile_only", **kw)
            assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (got - ref).abs().max().item()
            for r in range(reps):
                want = me.conv_forward(x[r * m:(r + 1) * m].double(), (w if k > 1 else w[0]).double(), nbr_np)
                if epi & 1:
                    want = want * sc.double() + sh.double()
                if epi & 2:
                    want = torch.relu(want + res[r * m:(r + 1) * m].double())
                assert torch.allclose(got[r * m:(r + 1) * m].cpu().double(), want, rtol=RTOL, atol=ATOL), (cin, k, r)


@pytest.mark.parametrize("cin,split,cout", [(64, 0, 64), (32, 0, 64), (96, 64, 64)])
def test_spconv_256_row_tiles_vs_128_row_tiles_and_oracle(device, cin, split, cout):
    """64-column layers on maps of >= 131 072 rows (replicas included) run on 256-row tiles (same accumulator LDS as a
    128 x 128 tile, twice the pairs per offset and stage): the 128-row kernel's results up to the order in which a tile adds
    its offsets (a tile decides from its own pair counts whether it packs several offsets into a stage, and packed stages
    add in a different fixed order: 1 ulp), run-to-run identical, on a stride-4-like map (~6 neighbours per voxel), a
    low-density one (packed stages: more than 8 segments per offset possible) and a ragged last tile; against the oracle
    on a sample of rows."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(7 + cin)
    for cloud, reps, hint in ((random_cloud(200000, 42, 61), 2, False), (random_cloud(80000, 90, 62, batch=2), 2, True),
                              (random_cloud(140000, 30, 63), 2, False)):
        uniq, _, _ = me.voxelize(cloud)
        nbr_np = me.kernel_map(uniq, uniq, 3, 1)
        nbr = dev_i32(nbr_np, device)
        m = uniq.shape[0]
        assert m * reps >= 256 * 512, m
        x = torch.randn(reps * m, cin, generator=g)
        w = torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = torch.randn(reps * m, cout, generator=g)
        xd = x.to(device)
        a = xd[:, :split].contiguous() if split else xd
        kw = dict(in_b=xd[:, split:].contiguous() if split else None, scale=sc.to(device), shift=sh.to(device),
                  residual=res.to(device), relu=True, replicas=reps, sparse_map=hint)
        got = ops.spconv_fwd(a, w.to(device), nbr, m, **kw)
        ref = ops.spconv_fwd(a, w.to(device), nbr, m, kernel="tile128", **kw)
        assert torch.equal(got, ops.spconv_fwd(a, w.to(device), nbr, m, **kw))
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5), (cin, cout, m, (got - ref).abs().max().item())
        rows = np.random.default_rng(5).choice(m, 4000, replace=False)
        sub = nbr_np[:, rows]
        want = me.conv_forward(x[:m].double(), w.double(), sub)
        want = torch.relu(want * sc.double() + sh.double() + res[:m][rows].double())
        assert torch.allclose(got[:m][rows].cpu().double(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("pieces", [3, 2])
@pytest.mark.parametrize("cin,split,cout", [(128, 0, 128), (256, 0, 256), (384, 256, 256), (128, 0, 256), (192, 128, 128), (64, 0, 128),
                                            (64, 0, 64), (32, 0, 64), (96, 64, 192)])
def test_spconv_split3_is_an_fp32_convolution(device, cin, split, cout, pieces):
    """lidiff_spconv_fwd_split3 (round 6): fp32 in / fp32 out with the contraction on the bf16 matrix pipe from three-way split
    operands -- against the float64 oracle at the NATIVE fp32 kernel's own bars (RTOL / ATOL of this file), and its worst error
    next to the native kernel's on the same inputs (recorded; asserted within 1.5x + one ulp of the output scale): dense and
    nearly empty maps, tiles without a pair, the CFG pair stacked, a ragged last tile, fused ME.cat, every epilogue, a
    device-side row count, the identity map (kernel_size 1), and the pieces of the OUTPUT written by the epilogue (their sum is
    the fp32 output, bit for bit).  pieces = 2: the opt-in mode on two fp16 pieces per operand (22-bit operands, three products) --
    the SAME bars, the same checks (its pieces sum to the output to 2^-21)."""
    from lidiff_amd import ops
    with ops.split_pieces(pieces):
        _split_conv_case(device, cin, split, cout, pieces)
    ops.split_check()


def _split_conv_case(device, cin, split, cout, pieces):
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(5 + cin + cout)
    worst = {}
    for n_pts, extent in ((60000, 14), (3000, 40), (130, 3)):
        cloud = random_cloud(n_pts, extent, 23 + n_pts)
        uniq, _, _ = me.voxelize(cloud)
        nbr_np = me.kernel_map(uniq, uniq, 3, 1)
        nbr = dev_i32(nbr_np, device)
        m = uniq.shape[0]
        for reps in (1, 2):
            x = torch.randn(reps * m, cin, generator=g) * (torch.rand(cin, generator=g) * 4 + 0.05)      # mixed channel scales
            w = torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)
            sc, sh, res = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g), torch.randn(reps * m, cout, generator=g)
            xd = x.to(device)
            a = xd[:, :split].contiguous() if split else xd
            b = xd[:, split:].contiguous() if split else None
            kw = dict(scale=sc.to(device), shift=sh.to(device), residual=res.to(device), relu=True, replicas=reps)
            got = ops.spconv_fwd_split3(a, w.to(device), nbr, m, in_b=b, want_planes=True, **kw)
            assert torch.equal(got, ops.spconv_fwd_split3(a, w.to(device), nbr, m, in_b=b, **kw))          # run to run
            native = ops.spconv_fwd(a, w.to(device), nbr, m, in_b=b, **kw)
            pick = np.random.default_rng(5).choice(m, min(m, 3000), replace=False)
            for r in range(reps):
                want = me.conv_forward(x[r * m:(r + 1) * m].double(), w.double(), nbr_np[:, pick])
                want = torch.relu(want * sc.double() + sh.double() + res[r * m:(r + 1) * m][pick].double())
                e3 = (got[r * m:(r + 1) * m][pick].cpu().double() - want).abs().max().item()
                en = (native[r * m:(r + 1) * m][pick].cpu().double() - want).abs().max().item()
                assert torch.allclose(got[r * m:(r + 1) * m][pick].cpu().double(), want, rtol=RTOL, atol=ATOL)
                scale_out = want.abs().max().item()
                assert e3 <= 1.5 * en + 1.2e-7 * scale_out, (cin, cout, m, e3, en)
                worst[n_pts] = max(worst.get(n_pts, (0, 0)), (e3, en))
            # rows sorted by their neighbour sets (whole 16-row blocks then lack an offset and are skipped): the same bits -- a
            # skipped block contributes what the multiplied zeros contributed, nothing
            nbr_s, order = ops.mask_sorted_map(nbr)
            assert torch.equal(torch.sort(order.long()).values, torch.arange(m, device=device))
            got_s = ops.spconv_fwd_split3(a, w.to(device), nbr_s, m, in_b=b, want_planes=True, row_order=order, **kw)
            assert torch.equal(got_s, got) and torch.equal(got_s._lidiff_split3[1], got._lidiff_split3[1])
            planes = got._lidiff_split3[1].float()
            if pieces == 3:
                assert torch.equal(planes[:, 0] + planes[:, 1] + planes[:, 2], got)
            else:
                assert planes.shape[1] == 2 and bool(((planes[:, 0] + planes[:, 1] - got).abs() <= got.abs() * 2.0 ** -21 + 2.0 ** -25).all())
            assert torch.equal(planes, ops.split3_rows(got.clone()).float())
            # no epilogue
            plain = ops.spconv_fwd_split3(a, w.to(device), nbr, m, in_b=b, replicas=reps)
            ref = ops.spconv_fwd(a, w.to(device), nbr, m, in_b=b, replicas=reps)
            assert torch.allclose(plain, ref, rtol=1e-5, atol=1e-5)
            if m > 300:
                rows = torch.tensor([m - 77], dtype=torch.int32, device=device)
                nbr_d = nbr.clone()
                nbr_d[:, m - 77:] = -1
                got_d = ops.spconv_fwd_split3(a, w.to(device), nbr_d, m, in_b=b, d_rows=rows, **kw)
                for r in range(reps):
                    assert torch.equal(got_d[r * m:r * m + m - 77], got[r * m:r * m + m - 77])
                nbr_ds, order_d = ops.mask_sorted_map(nbr_d)               # (valid rows first: the count still describes them)
                assert int(order_d[:m - 77].max()) < m - 77
                got_ds = ops.spconv_fwd_split3(a, w.to(device), nbr_ds, m, in_b=b, d_rows=rows, row_order=order_d, **kw)
                for r in range(reps):
                    assert torch.equal(got_ds[r * m:r * m + m - 77], got[r * m:r * m + m - 77])
    # the transposed stride-2 map (8 offsets, ONE neighbour per output row: the decoder's up-convolutions): rows grouped by offset
    # (the mask sort of a one-neighbour table) == table order, bit for bit; against the oracle at the same bars
    coarse, _ = me.stride_map(uniq, 2)
    up_np = me.transpose_kernel_map(me.kernel_map(uniq, coarse, 2, 1), m)
    mc = coarse.shape[0]
    xc = torch.randn(reps * mc, cin, generator=g)
    w8 = torch.randn(8, cin, cout, generator=g) / np.sqrt(cin)
    up = dev_i32(up_np, device)
    ac = xc.to(device)[:, :split].contiguous() if split else xc.to(device)
    bc = xc.to(device)[:, split:].contiguous() if split else None
    got8 = ops.spconv_fwd_split3(ac, w8.to(device), up, m, in_b=bc, replicas=reps, relu=True)
    up_s, order8 = ops.mask_sorted_map(up)
    assert torch.equal(ops.spconv_fwd_split3(ac, w8.to(device), up_s, m, in_b=bc, replicas=reps, relu=True, row_order=order8), got8)
    for r in range(reps):
        want8 = torch.relu(me.conv_forward(xc[r * mc:(r + 1) * mc].double(), w8.double(), up_np))
        assert torch.allclose(got8[r * m:(r + 1) * m].cpu().double(), want8, rtol=RTOL, atol=ATOL)
    # kernel_size 1 (identity map)
    w1 = torch.randn(1, cin, cout, generator=g) / np.sqrt(cin)
    got1 = ops.spconv_fwd_split3(a, w1.to(device), None, m, in_b=b, replicas=reps)
    want1 = x.double() @ w1[0].double()
    assert torch.allclose(got1.cpu().double(), want1, rtol=RTOL, atol=ATOL)
    record_parity(f"split3_{cin}_{cout}" if pieces == 3 else f"split_f16x2_{cin}_{cout}", **{f"err_split3_vs_native_{k}": v for k, v in worst.items()})


def test_two_piece_fp16_split_reports_values_beyond_its_range(device):
    """The opt-in two-piece fp16 mode holds |x| <= 65504: a feature beyond that (or Inf / NaN) raises LIDIFF_STATUS_F16_RANGE on the
    device and ops.split_check() turns it into an exception -- from the row cut and from the epilogue's cut of an output alike;
    the default three-piece bf16 mode has fp32's range and reports nothing."""
    from lidiff_amd import ops
    x = torch.randn(4096, 64, device=device)
    w = torch.randn(1, 64, 64, device=device) / 8
    with ops.split_pieces(2):
        ops.spconv_fwd_split3(x, w, None, 4096, want_planes=True)
        ops.split_check()                                                   # in range: silent
        bad = x.clone()
        bad[17, 3] = 7.0e4
        ops.spconv_fwd_split3(bad, w, None, 4096)
        with pytest.raises(RuntimeError, match="fp16's range"):
            ops.split_check()
        ops.split_check()                                                   # (the word was cleared)
        big = ops.spconv_fwd_split3(x, w * 3.0e4, None, 4096, want_planes=True)      # outputs of ~1e5: the epilogue's cut overflows
        assert float(big.abs().max()) > 65504
        with pytest.raises(RuntimeError, match="fp16's range"):
            ops.split_check()
    with ops.split_pieces(3):
        got = ops.spconv_fwd_split3(bad, w, None, 4096, want_planes=True)
        ops.split_check()
    assert bool(torch.isfinite(got).all())


@pytest.mark.parametrize("m,c", [(2, 32), (37, 96), (5000, 256), (120001, 64), (70000, 128), (9, 4)])
def test_batch_norm_train_kernels_vs_torch_float64(device, m, c):
    """lidiff_bn_stats / lidiff_bn_apply / lidiff_bn_bwd (ops._BatchNormTrain: the training-mode MinkowskiBatchNorm) against
    nn.BatchNorm1d in float64 on the CPU: output, input / weight / bias gradients, running estimates and the batch counter;
    with a large common offset on the inputs (the variance must not cancel), with and without the fused ReLU, run-to-run
    identical."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(m + c)
    x = torch.randn(m, c, generator=g) * (torch.rand(c, generator=g) * 3 + 0.1) + torch.randn(c, generator=g) * 20
    r = torch.randn(m, c, generator=g)
    for relu in (False, True):
        ref = torch.nn.BatchNorm1d(c).double()
        with torch.no_grad():
            ref.weight.copy_(torch.rand(c, generator=g) + 0.5)
            ref.bias.copy_(torch.randn(c, generator=g))
        bn = torch.nn.BatchNorm1d(c).to(device)
        bn.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        ref.train(); bn.train()
        xr = x.double().requires_grad_(True)
        pre = ref(xr)
        yr = torch.relu(pre) if relu else pre
        (yr * r.double()).sum().backward()
        xd = x.to(device).requires_grad_(True)
        y = ops.batch_norm_train(xd, bn, relu=relu)
        (y * r.to(device)).sum().backward()
        scale = float(yr.abs().max())
        assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-5, atol=(2e-6 if m >= 1000 else 2e-4) * max(1.0, scale)), (m, c, relu)
        gs = float(xr.grad.abs().max())
        # (a handful of rows: nearly equal samples give a tiny variance, and dx is a difference of terms 1 / sqrt(var + eps)
        # times larger than itself -- float32 cancellation, not an error of the kernel: looser bar)
        g_rtol, g_atol = (1e-4, 2e-6 * max(1.0, gs)) if m >= 1000 else (2e-2, 2e-3 * max(1.0, gs))
        got_g, want_g = xd.grad.cpu().double(), xr.grad
        if relu:            # an output within rounding of the kink may fall on the other side of it: such elements carry a full dy
            flip = (pre.detach().abs() < 1e-5 * max(1.0, scale))
            assert flip.float().mean() < 1e-3
            far = ~flip.any(1)                           # (a flipped element changes its whole row's dx through the sums only weakly)
            assert torch.allclose(got_g[far], want_g[far], rtol=max(g_rtol, 1e-3), atol=max(g_atol, 1e-4 * gs)), (m, c, relu)
        else:
            assert torch.allclose(got_g, want_g, rtol=g_rtol, atol=g_atol), (m, c, relu)
        p_tol = 1e-4 if not relu else 1e-2          # (with ReLU a few outputs at the kink change side: whole dy terms in the sums)
        assert torch.allclose(bn.weight.grad.cpu().double(), ref.weight.grad, rtol=p_tol, atol=p_tol * float(ref.weight.grad.abs().max()))
        assert torch.allclose(bn.bias.grad.cpu().double(), ref.bias.grad, rtol=p_tol, atol=p_tol * float(ref.bias.grad.abs().max()))
        assert torch.allclose(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn.num_batches_tracked) == 1
        bn2 = torch.nn.BatchNorm1d(c).to(device)
        bn2.load_state_dict({k: v.float() for k, v in ref.state_dict().items() if "running" not in k and "num" not in k}, strict=False)
        y2 = ops.batch_norm_train(x.to(device), bn2, relu=relu)
        assert torch.equal(y2, y.detach())
        # the ResidualBlock form: relu?(bn(x) + residual), the residual's gradient = dy behind the ReLU
        res = torch.randn(m, c, generator=g)
        ref3 = torch.nn.BatchNorm1d(c).double()
        ref3.load_state_dict({k: v for k, v in ref.state_dict().items() if "running" not in k and "num" not in k}, strict=False)
        bn3 = torch.nn.BatchNorm1d(c).to(device)
        bn3.load_state_dict({k: v.float() for k, v in ref3.state_dict().items()})
        x3, r3 = x.double().requires_grad_(True), res.double().requires_grad_(True)
        pre3 = ref3(x3) + r3
        y3r = torch.relu(pre3) if relu else pre3
        (y3r * r.double()).sum().backward()
        xd3, rd3 = x.to(device).requires_grad_(True), res.to(device).requires_grad_(True)
        y3 = ops.batch_norm_train(xd3, bn3, relu=relu, residual=rd3)
        (y3 * r.to(device)).sum().backward()
        y_atol = (2e-6 if m >= 1000 else 2e-4) * max(1.0, float(y3r.abs().max()))    # (few rows: float32 mean of offset data)
        assert torch.allclose(y3.detach().cpu().double(), y3r.detach(), rtol=1e-5, atol=y_atol)
        keep = (pre3.detach().abs() >= 1e-5 * max(1.0, float(pre3.abs().max()))) if relu else torch.ones_like(pre3, dtype=torch.bool)
        assert torch.allclose(rd3.grad.cpu().double()[keep], r3.grad[keep], rtol=1e-6, atol=1e-7)
        if m >= 1000 and not relu:
            assert torch.allclose(xd3.grad.cpu().double(), x3.grad, rtol=1e-4, atol=2e-6 * max(1.0, float(x3.grad.abs().max())))


def test_scatter_add_as_segment_sum_is_deterministic(device):
    """ops.scatter_add_rows through lidiff_segment_sum_rows (sources stably sorted by destination, CSR, no atomics): equal to a
    float64 index_add, identical from run to run, for a many-to-one index (a match index), a near-permutation (an inverse
    mapping), destinations without sources, and widths that are not multiples of 4."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(4)
    # ... and destinations with very many sources -- the unconditional training branch gathers ~180 000 rows from each
    # of 2 part voxels -- which go through the worklist to the cooperative long-segment kernel: (180000, 2, 256), all-long
    # (100000, 700, 96), long + short mixed with an odd width (5000, 40, 10)
    for n, m, c in ((200000, 6000, 96), (50000, 49000, 3), (1000, 5, 32), (7, 20, 4), (180000, 2, 256), (100000, 700, 96),
                    (5000, 40, 10)):
        idx = torch.randint(0, m, (n,), generator=g)
        if m == 40:
            idx = torch.where(idx < 20, torch.zeros_like(idx), idx)      # destination 0 takes half the sources, 1 .. 19 none
        if m == 20:
            idx = idx.clamp(max=9)                       # destinations 10 .. 19 stay empty
        src = torch.randn(n, c, generator=g) * 3
        want = torch.zeros(m, c, dtype=torch.float64).index_add_(0, idx, src.double())
        got = ops.scatter_add_rows(src.to(device), idx.to(device), m)
        assert torch.equal(got, ops.scatter_add_rows(src.to(device), idx.to(device), m))
        # (fp32 sums of n / m terms of magnitude ~3: the rounding error grows with the segment length -- 1e-3 measured on the
        # 90 000-term segments)
        tol = 1e-4 + 6e-7 * n / m
        assert torch.allclose(got.cpu().double(), want, rtol=1e-5, atol=tol)
    # the cliff itself: B = 2 x 180 000 rows onto 2 destinations, 256 channels -- a thread per (destination, float4) walking
    # its segment took tens of ms; the cooperative kernel is bounded by reading the rows once
    n, c = 360000, 256
    idx = (torch.arange(n) >= n // 2).long().to(device)
    src = torch.randn(n, c, generator=g).to(device)
    ops.scatter_add_rows(src, idx, 2)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    out = ops.scatter_add_rows(src, idx, 2)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    record_parity("segment_sum_long_segments", ms_360k_rows_onto_2=ms)
    assert torch.allclose(out.cpu().double(), torch.stack([src[:n // 2].double().sum(0), src[n // 2:].double().sum(0)]).cpu(), rtol=1e-4, atol=2e-2)
    # (the time is recorded, not asserted: a wall-clock bound is no correctness test -- bench.py --layer-table times kernels)


ROW_KERNEL_SHAPES = [(32, 0, 32), (32, 0, 64), (64, 0, 64), (64, 0, 128), (96, 0, 96), (96, 64, 96), (128, 96, 96),
                     (128, 0, 128), (128, 0, 256), (192, 128, 128), (64, 32, 32), (192, 0, 96)]


@pytest.mark.parametrize("cin,split,cout", ROW_KERNEL_SHAPES)
def test_spconv_row_kernel_is_bit_identical_to_the_tile_kernel(device, cin, split, cout):
    """spconv_rows.hip (identity maps as a streaming row GEMM: W tile in LDS, rows from HBM straight into the MFMA
    operands, transposed product, 16-byte stores) on every input width it accepts -- row counts of 1, 15, 16, 17 and
    thousands (ragged last block, fewer blocks than waves, several blocks per wave), fused ME.cat, replicas 1 / 2, every
    epilogue combination, the centre pass with tail rows through the CSR -- bit for bit against the tile kernel (same MFMA
    sequence per output, same tail order) and against the float64 oracle."""
    from lidiff_amd import _lib, ops
    g = torch.Generator().manual_seed(cin * 7 + cout)
    lib = _lib.load()
    assert lib.lidiff_spconv_fwd_kernel_id(split or cin, cin - split if split else 0, cout, 1, 0, 0, 0) == 2
    assert lib.lidiff_spconv_fwd_kernel_id(split or cin, cin - split if split else 0, cout, 1, 0, 0, 8) == 0      # TILE_ONLY
    assert lib.lidiff_spconv_fwd_kernel_id(split or cin, cin - split if split else 0, cout, 1, 0, 1, 0) == 0      # row order
    assert lib.lidiff_spconv_fwd_kernel_id(split or cin, cin - split if split else 0, cout, 27, 1, 0, 0) == 0     # a real map
    w = torch.randn(cin, cout, generator=g) / np.sqrt(cin)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    for m, reps, epi in ((1, 1, 0), (15, 2, 1), (16, 1, 2), (17, 2, 3), (4099, 2, 3), (70001, 1, 3), (33333, 2, 0)):
        x = torch.randn(reps * m, cin, generator=g)
        res = torch.randn(reps * m, cout, generator=g)
        xd = x.to(device)
        a = xd[:, :split].contiguous() if split else xd
        kw = dict(in_b=xd[:, split:].contiguous() if split else None, replicas=reps,
                  scale=sc.to(device) if epi & 1 else None, shift=sh.to(device) if epi & 1 else None,
                  residual=res.to(device) if epi & 2 else None, relu=bool(epi & 2))
        got = ops.spconv_fwd(a, w.to(device), None, m, **kw)
        ref = ops.spconv_fwd(a, w.to(device), None, m, kernel="tile_only", **kw)
        assert torch.equal(got, ref), (m, reps, epi, (got - ref).abs().max().item())
        want = x.double() @ w.double()
        if epi & 1:
            want = want * sc.double() + sh.double()
        if epi & 2:
            want = torch.relu(want + res.double())
        assert torch.allclose(got.cpu().double(), want, rtol=RTOL, atol=ATOL), (m, reps, epi)
    # the centre pass of a kernel_size-3 map with its tail rows
    for cloud in (random_cloud(6000, 40, 11, batch=2, dup=0.05), random_cloud(1500, 4, 13)):
        uniq, _, _ = me.voxelize(cloud)
        nbr = dev_i32(me.kernel_map(uniq, uniq, 3, 1), device)
        m = uniq.shape[0]
        tmap = ops.TailMap(nbr)
        assert tmap.n > 0
        w3 = (torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)).to(device)
        x = torch.randn(2 * m, cin, generator=g).to(device)
        a = x[:, :split].contiguous() if split else x
        in_b = x[:, split:].contiguous() if split else None
        rows = ops.spconv_fwd(a, w3, tmap.nbr, tmap.n, in_b=in_b, replicas=2)
        kw = dict(in_b=in_b, replicas=2, scale=sc.to(device), shift=sh.to(device), relu=True, tail=(rows, tmap.ptr, tmap.idx),
                  offset=13)
        got = ops.spconv_fwd(a, w3, None, m, **kw)
        assert torch.equal(got, ops.spconv_fwd(a, w3, None, m, kernel="tile_only", **kw))


@pytest.mark.parametrize("cin,split,cout", [(32, 0, 32), (64, 0, 64), (96, 0, 96), (128, 96, 96), (128, 0, 128), (64, 32, 128),
                                            (128, 0, 256)])
def test_spconv_pair_list_kernel_is_bit_identical_to_the_tile_kernel(device, cin, split, cout):
    """lidiff_spconv_fwd_pairs (spconv_rows.hip, gathered rows, one W tile per kernel offset) on the two map kinds it serves
    -- the transposed kernel_size-2 / stride-2 map as its rulebook (every fine voxel one pair; uneven, partly EMPTY offsets
    included) and the tail-pass map of centre + tail (one output row per pair) -- bit for bit against the tile kernel over
    the same map (plain and offset-grouped row order), with fused ME.cat, epilogue, residual and replicas 1 / 2, and against
    the float64 oracle."""
    from lidiff_amd import _lib, ops
    assert _lib.load().lidiff_spconv_fwd_pairs_supported(split or cin, cin - split if split else 0, cout) == 1
    assert _lib.load().lidiff_spconv_fwd_pairs_supported(192, 0, 128) == 0 and _lib.load().lidiff_spconv_fwd_pairs_supported(3, 0, 32) == 0
    g = torch.Generator().manual_seed(cin + 3 * cout)
    st = status(device)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(device), torch.randn(cout, generator=g).to(device)
    for cloud, reps in ((random_cloud(9000, 30, 31, batch=2), 2), (random_cloud(5000, 6, 32), 1), (random_cloud(20, 40, 33), 2)):
        fine, _, _, _ = ops.vox_unique(dev_i32(cloud, device), st)
        coarse, parent, _ = ops.map_stride(fine, 2, st)
        up = ops.kernel_map_up(fine, parent, 1)                        # [8, m_fine]: input rows are COARSE voxels
        m_in, m_out = coarse.shape[0], fine.shape[0]
        pin, pout, off = ops.rulebook_compact(up, total=m_out)
        w = (torch.randn(8, cin, cout, generator=g) / np.sqrt(cin)).to(device)
        x = torch.randn(reps * m_in, cin, generator=g).to(device)
        res = torch.randn(reps * m_out, cout, generator=g).to(device)
        a = x[:, :split].contiguous() if split else x
        kw = dict(in_b=x[:, split:].contiguous() if split else None, scale=sc, shift=sh, residual=res, relu=True, replicas=reps)
        got = ops.spconv_fwd_pairs(a, w, pin, pout, off, m_out, **kw)
        ref = ops.spconv_fwd(a, w, up, m_out, sparse_map=True, **kw)
        assert torch.equal(got, ref), (cin, cout, m_out, (got - ref).abs().max().item())
        grouped = ops.spconv_fwd(a, w, up.index_select(1, pout.long()).contiguous(), m_out, row_order=pout, **kw)
        assert torch.equal(got, grouped)
        up_np = up.cpu().numpy()
        want = me.conv_forward(x[:m_in].cpu().double(), w.cpu().double(), up_np)
        want = torch.relu(want * sc.cpu().double() + sh.cpu().double() + res[:m_out].cpu().double())
        assert torch.allclose(got[:m_out].cpu().double(), want, rtol=RTOL, atol=ATOL)
    # a map whose offsets 1..7 are all empty: every fine voxel sits in the corner of its parent cell
    even = random_cloud(3000, 20, 34)
    even[:, 1:] *= 2
    fine, _, _, _ = ops.vox_unique(dev_i32(even, device), st)
    coarse, parent, _ = ops.map_stride(fine, 2, st)
    up = ops.kernel_map_up(fine, parent, 1)
    pin, pout, off = ops.rulebook_compact(up, total=fine.shape[0])
    assert int((off[1:] - off[:-1] > 0).sum()) == 1
    w = (torch.randn(8, cin, cout, generator=g) / np.sqrt(cin)).to(device)
    x = torch.randn(coarse.shape[0], cin, generator=g).to(device)
    a = x[:, :split].contiguous() if split else x
    in_b = x[:, split:].contiguous() if split else None
    assert torch.equal(ops.spconv_fwd_pairs(a, w, pin, pout, off, fine.shape[0], in_b=in_b),
                       ops.spconv_fwd(a, w, up, fine.shape[0], in_b=in_b))
    # the tail pass: one output row per pair, 26 uneven offsets
    for cloud in (random_cloud(6000, 40, 11, batch=2, dup=0.05), random_cloud(1500, 4, 13)):
        uniq, _, _ = me.voxelize(cloud)
        nbr = dev_i32(me.kernel_map(uniq, uniq, 3, 1), device)
        tmap = ops.TailMap(nbr)
        w3 = (torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)).to(device)
        x = torch.randn(2 * uniq.shape[0], cin, generator=g).to(device)
        a = x[:, :split].contiguous() if split else x
        in_b = x[:, split:].contiguous() if split else None
        assert torch.equal(tmap.pair_in, tmap.nbr.amax(0)) and int(tmap.off[-1]) == tmap.n
        got = ops.spconv_fwd_pairs(a, w3, tmap.pair_in, None, tmap.off, tmap.n, in_b=in_b, replicas=2)
        assert torch.equal(got, ops.spconv_fwd(a, w3, tmap.nbr, tmap.n, in_b=in_b, replicas=2))
    assert int(st.item()) == 0


@pytest.mark.parametrize("cin,cout,split", [(32, 32, 0), (96, 96, 0), (128, 96, 96), (3, 32, 0), (64, 128, 0)])
def test_spconv_centre_tail_vs_oracle(device, cin, cout, split):
    """Low-density kernel_size-3 maps as centre pass + tail rows (ops.TailMap / spconv_centre_tail): the pairs of the 26
    non-centre offsets multiplied offset by offset into one row per pair, added through the CSR of the map in the epilogue
    of the dense centre pass -- against the oracle (rtol / atol 1e-4), against the one-launch kernel, run-to-run identical,
    with epilogue, fused ME.cat and replicas; a map with NO non-centre pair and a dense map included."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    clouds = [random_cloud(6000, 40, 11, batch=2, dup=0.05), random_cloud(300, 200, 12, dup=0.0), random_cloud(1500, 4, 13)]
    for cloud in clouds:
        uniq, _, _ = me.voxelize(cloud)
        nbr_np = me.kernel_map(uniq, uniq, 3, 1)
        nbr = dev_i32(nbr_np, device)
        m = uniq.shape[0]
        tmap = ops.TailMap(nbr)
        assert tmap.n == int((nbr_np >= 0).sum()) - m
        if tmap.n:                                              # CSR: every pair once, under its output row, ascending offset
            ptr, idx = tmap.ptr.cpu().numpy(), tmap.idx.cpu().numpy()
            assert ptr[0] == 0 and ptr[-1] == tmap.n and sorted(idx.tolist()) == list(range(tmap.n))
            tail_np = tmap.nbr.cpu().numpy()
            k_of = tail_np.argmax(0)
            assert np.all((tail_np >= 0).sum(0) == 1)
            for o in (0, m // 2, m - 1):
                ks = k_of[idx[ptr[o]:ptr[o + 1]]]
                want_k = [k for k in range(27) if k != 13 and nbr_np[k, o] >= 0]
                assert ks.tolist() == want_k
                assert tail_np[ks, idx[ptr[o]:ptr[o + 1]]].tolist() == [nbr_np[k, o] for k in want_k]
        x = torch.randn(2 * m, cin, generator=g)
        w = torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 9)
        sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = torch.randn(2 * m, cout, generator=g)
        xd = x.to(device)
        a = xd[:, :split].contiguous() if split else xd
        kw = dict(in_b=xd[:, split:].contiguous() if split else None, scale=sc.to(device), shift=sh.to(device),
                  residual=res.to(device), relu=True, replicas=2)
        got = ops.spconv_centre_tail(a, w.to(device), tmap, m, **kw)
        assert torch.equal(got, ops.spconv_centre_tail(a, w.to(device), tmap, m, **kw))
        one = ops.spconv_fwd(a, w.to(device), nbr, m, sparse_map=True, **kw)
        assert torch.allclose(got, one, rtol=1e-4, atol=1e-4)
        for r in range(2):
            want = me.conv_forward(x[r * m:(r + 1) * m].double(), w.double(), nbr_np)
            want = torch.relu(want * sc.double() + sh.double() + res[r * m:(r + 1) * m].double())
            assert torch.allclose(got[r * m:(r + 1) * m].cpu().double(), want, rtol=RTOL, atol=ATOL), (cin, cout, r)


def test_voxel_mean_and_slice_backward_vs_autograd(device):
    """The backward of TensorField.sparse() (UNWEIGHTED_AVERAGE: lidiff_vox_mean_bwd) and of SparseTensor.slice
    (scatter-add of the point gradients into the voxel rows) against torch autograd through the oracle's formulation."""
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    coords = random_cloud(5000, 8, 77, batch=2)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(coords.shape[0], 3, generator=g)
    r = torch.randn(coords.shape[0], 7, generator=g)
    fd = feats.to(device).requires_grad_(True)
    field = ME.TensorField(features=fd, coordinates=torch.from_numpy(coords).float().to(device), device=device)
    sp = field.sparse()
    w = torch.randn(3, 7, generator=g)
    y = ME.SparseTensor(sp.F @ w.to(device), tensor_stride=1, coordinate_manager=sp.coordinate_manager).slice(field).F
    (y * r.to(device)).sum().backward()
    uniq, inv, _ = me.voxelize(coords)
    fo = feats.clone().requires_grad_(True)
    vox = me.voxel_mean(fo, inv, uniq.shape[0])
    yo = (vox @ w)[torch.from_numpy(inv)]
    (yo * r).sum().backward()
    assert torch.allclose(y.detach().cpu(), yo.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(fd.grad.cpu(), fo.grad, rtol=1e-4, atol=1e-5)
    # the two kernels directly
    gv = torch.randn(uniq.shape[0], 3, generator=g)
    counts = torch.from_numpy(np.bincount(inv, minlength=uniq.shape[0]).astype(np.float32))
    got = ops.vox_mean_bwd(gv.to(device), torch.from_numpy(inv).to(device), counts.to(device)).cpu()
    assert torch.allclose(got, gv[torch.from_numpy(inv)] / counts[torch.from_numpy(inv)][:, None], rtol=1e-6, atol=1e-7)


def _bf16r(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("kind,cin,cout,split,epilogue", [
    ("k3", 32, 32, 0, False), ("k3", 32, 64, 0, True), ("k3", 64, 128, 0, False), ("k3", 96, 96, 0, True),
    ("k3", 128, 256, 0, False), ("k3", 384, 256, 256, True), ("k3", 192, 128, 128, False), ("k3", 160, 96, 96, False),
    ("down", 64, 64, 0, False), ("up", 256, 256, 0, False), ("up", 128, 96, 0, True), ("k1", 96, 32, 0, False)])
def test_spconv_bf16_operands_vs_oracle_on_rounded_inputs(device, kind, cin, cout, split, epilogue):
    """lidiff_spconv_fwd_bf16 (BASELINE configs[4], bf16 training): bf16 operands, fp32 accumulation.  The oracle convolves
    the bf16-ROUNDED inputs and weights in float64, so the only difference left is the fp32 summation order -- the bar is
    the fp32 kernel's own (rtol 1e-4 / atol 1e-4), not a bf16-sized tolerance."""
    from lidiff_amd import ops
    coords = random_cloud(4000, 6, 40, batch=2)
    uniq, _, _ = me.voxelize(coords)
    coarse, _ = me.stride_map(uniq, 2)
    if kind == "k3":
        nbr, m_in, m_out, K = me.kernel_map(uniq, uniq, 3, 1), uniq.shape[0], uniq.shape[0], 27
    elif kind == "down":
        nbr, m_in, m_out, K = me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0], coarse.shape[0], 8
    elif kind == "up":
        nbr = me.transpose_kernel_map(me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0])
        m_in, m_out, K = coarse.shape[0], uniq.shape[0], 8
    else:
        nbr, m_in, m_out, K = None, uniq.shape[0], uniq.shape[0], 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn(m_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) / np.sqrt(cin * max(1, K // 3))
    want = me.conv_forward(_bf16r(x).double(), _bf16r(w).double() if K > 1 else _bf16r(w)[0].double(), nbr)
    scale = shift = res = None
    if epilogue:
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        res = torch.randn(m_out, cout, generator=g)
        want = torch.relu(want * scale.double() + shift.double() + res.double())
    d = lambda t: None if t is None else t.to(device)
    nbr_d = None if nbr is None else dev_i32(nbr, device)
    kw = dict(scale=d(scale), shift=d(shift), residual=d(res), relu=epilogue)
    if split:
        got = ops.spconv_fwd_bf16(d(x[:, :split].contiguous()), d(w), nbr_d, m_out, in_b=d(x[:, split:].contiguous()), **kw)
    else:
        got = ops.spconv_fwd_bf16(d(x), d(w), nbr_d, m_out, **kw)
    torch.cuda.synchronize()
    err = (got.cpu().double() - want).abs().max().item()
    assert torch.allclose(got.cpu().double(), want, rtol=RTOL, atol=ATOL), f"{kind} {cin}->{cout}: max err {err}"
    # ... and it is the fp32 kernel's answer on the rounded operands, and NOT its answer on the unrounded ones
    f32 = ops.spconv_fwd(d(_bf16r(x)), d(_bf16r(w)), nbr_d, m_out, **kw) if not split else None
    if f32 is not None:
        assert torch.allclose(got, f32, rtol=RTOL, atol=ATOL)
    # replicas share the map and the weights
    if not split and not epilogue:
        x2 = torch.cat([x, -2.0 * x])
        got2 = ops.spconv_fwd_bf16(d(x2), d(w), nbr_d, m_out, replicas=2)
        assert torch.equal(got2[:m_out], got)
        # (the matrix unit's internal sum is not sign-symmetric to the last bit, so the negated replica is compared closely, not exactly)
        assert torch.allclose(got2[m_out:], -2.0 * got, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("kind,ks,stride,cin,cout", [("conv", 3, 1, 32, 64), ("conv", 2, 2, 64, 64), ("tconv", 2, 2, 64, 32),
                                                    ("conv", 1, 1, 96, 32), ("conv", 3, 1, 256, 256), ("conv", 3, 1, 32, 48)])
def test_spconv_bf16_training_backward_vs_oracle_autograd(device, kind, ks, stride, cin, cout):
    """The ME-shim convolution under ops.train_operands("bf16"): forward and dX through lidiff_spconv_fwd_bf16 (dX over the
    swapped map with the transposed, flipped kernel), dW through lidiff_spconv_bwd_w_bf16.  The float64 oracle under
    me.bf16_operands() rounds the same operands to bf16 -- forward conv(bf16 x, bf16 W), dX conv^T(bf16 g, bf16 W),
    dW bf16(x)^T bf16(g) -- so it is the exact reference of all three up to fp32 summation order.  A layer whose channel
    counts are not multiples of 32 (32 -> 48 here) keeps the fp32 kernels under the same switch."""
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    coords = random_cloud(1500, 5, 31, batch=2)
    g = torch.Generator().manual_seed(7)
    uniq, _, _ = me.voxelize(coords)
    field = ME.TensorField(features=torch.randn(coords.shape[0], 3, generator=g).to(device),
                           coordinates=torch.from_numpy(coords).float().to(device), device=device)
    mgr = field.sparse().coordinate_manager
    ts_in = 1
    if kind == "tconv":
        mgr.stride(1, 2)
        ts_in = 2
    m_in = mgr.maps[ts_in].coords.shape[0]
    xf = torch.randn(m_in, cin, generator=g)
    x = ME.SparseTensor(xf.to(device).requires_grad_(True), tensor_stride=ts_in, coordinate_manager=mgr)
    mod = (ME.MinkowskiConvolutionTranspose if kind == "tconv" else ME.MinkowskiConvolution)(
        cin, cout, kernel_size=ks, stride=stride, dimension=3).to(device)
    takes_bf16 = ops.bf16_conv_applies(cin, 0, cout)
    prof = ops.ConvProfiler()
    ops.PROFILER = prof
    sparse_prev = ops.BF16_SPARSE_MAPS
    ops.BF16_SPARSE_MAPS = True                 # this small cloud's maps carry the sparse hint
    try:
        with ops.train_operands("bf16"):
            y = mod(x)
            r = torch.randn(y.F.shape, generator=g)
            (y.F * r.to(device)).sum().backward()
    finally:
        ops.PROFILER = None
        ops.BF16_SPARSE_MAPS = sparse_prev
    assert [v for v, *_ in prof.launches].count("bf16") == (2 if takes_bf16 else 0)
    coarse, _ = me.stride_map(uniq, 2)
    if kind == "tconv":
        nbr = me.transpose_kernel_map(me.kernel_map(uniq, coarse, 2, 1), uniq.shape[0])
    elif ks == 1:
        nbr = None
    elif stride == 2:
        nbr = me.kernel_map(uniq, coarse, 2, 1)
    else:
        nbr = me.kernel_map(uniq, uniq, 3, 1)
    w = mod.kernel.detach().cpu()
    xo = xf.double().requires_grad_(True)
    wo = w.double().requires_grad_(True)
    with me.bf16_operands():            # the oracle's emulation: operands rounded to bf16 where the channel counts allow
        yo = me.conv_forward(xo, wo, nbr)
    assert torch.allclose(y.F.detach().cpu().double(), yo.detach(), rtol=RTOL, atol=ATOL)
    (yo * r.double()).sum().backward()
    assert torch.allclose(x.F.grad.cpu().double(), xo.grad, rtol=1e-4, atol=1e-4), "dX"
    assert torch.allclose(mod.kernel.grad.cpu().double(), wo.grad, rtol=1e-4, atol=1e-3), "dW"


@pytest.mark.parametrize("c,reps,n,m", [(96, 2, 50000, 41000), (96, 1, 777, 300), (32, 2, 1, 1), (128, 3, 9000, 9000)])
def test_slice_and_head_in_one_launch_vs_torch_float64(device, c, reps, n, m):
    """lidiff_slice_head: head(feats[inverse]) for `self.last` = Linear(C, 20), LeakyReLU(0.1), Linear(20, 3) (minkunet.py:390,497),
    the replicas of a CFG pair stacked -- against the float64 evaluation of the same lines (bar 2e-6 absolute on O(1) outputs:
    fp32 sums of C and 20 terms), equal to torch's own fp32 gather + GEMMs within the same bar, and bit-identical when the voxel
    matrix is handed over with rows behind the ones the inverse map uses (a bound instead of the exact size)."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(3 * c + n)
    head = torch.nn.Sequential(torch.nn.Linear(c, 20), torch.nn.LeakyReLU(0.1, inplace=True), torch.nn.Linear(20, 3)).to(device)
    assert ops.slice_head_applies(head)
    feats = torch.randn(reps * m, c, generator=g).to(device)
    inv = torch.randint(0, m, (n,), generator=g).to(device)
    got = ops.slice_head(feats, inv, head, replicas=reps)
    assert got.shape == (reps * n, 3)
    rows = torch.cat([feats[r * m:(r + 1) * m][inv] for r in range(reps)])
    with torch.no_grad():
        want64 = head.double()(rows.double())
        head.float()
        want32 = head(rows.clone())
    assert torch.allclose(got.double(), want64, rtol=0, atol=2e-6), (got.double() - want64).abs().max().item()
    assert torch.allclose(got, want32, rtol=0, atol=2e-6)
    # the same rows inside a larger (bound-sized) voxel matrix per replica
    pad = torch.randn(reps, m + 1000, c, generator=g).to(device)
    for r in range(reps):
        pad[r, :m] = feats[r * m:(r + 1) * m]
    assert torch.equal(ops.slice_head(pad.reshape(-1, c), inv, head, replicas=reps), got)
    assert not ops.slice_head_applies(torch.nn.Sequential(torch.nn.Linear(c, 24), torch.nn.LeakyReLU(0.1), torch.nn.Linear(24, 3)).to(device))


@pytest.mark.parametrize("cin,split,cout,kind", [(256, 0, 256, "k3"), (128, 64, 128, "k3"), (96, 0, 96, "k3"), (32, 0, 64, "k3"),
                                                 (64, 0, 32, "down"), (160, 96, 96, "k3"), (64, 0, 64, "k1")])
def test_spconv_bf16_from_shadow_rows_is_bit_identical(device, cin, split, cout, kind):
    """lidiff_spconv_fwd_bf16 / lidiff_spconv_bwd_w_bf16 with in_bf16 = 1: the feature rows arrive as bf16 (lidiff_cast_bf16:
    round to nearest even -- the conversion the kernels otherwise apply to every gathered fp32 row).  Same operands, same
    products, same order of sums: the forward (every tile width, 64- / 32-channel stages, split inputs, epilogue, replicas),
    the transposed form the input gradient uses and the weight gradient equal the fp32-row form bit for bit."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(5 + cin + cout)
    uniq, _, _ = me.voxelize(random_cloud(9000, 14, 81, batch=2))
    if kind == "k3":
        nbr_np = me.kernel_map(uniq, uniq, 3, 1)
    elif kind == "down":
        coarse, _ = me.stride_map(uniq, 2)
        nbr_np = me.kernel_map(uniq, coarse, 2, 1)
    else:
        nbr_np = None
    m_in = uniq.shape[0]
    m_out = nbr_np.shape[1] if nbr_np is not None else m_in
    k = nbr_np.shape[0] if nbr_np is not None else 1
    nbr = dev_i32(nbr_np, device) if nbr_np is not None else None
    reps = 2
    x = torch.randn(reps * m_in, cin, generator=g).to(device)
    w = (torch.randn(k, cin, cout, generator=g) / np.sqrt(cin * k / 3)).to(device)
    xa = x[:, :split].contiguous() if split else x
    xb = x[:, split:].contiguous() if split else None
    x16 = ops.cast_bf16(x)
    assert x16.dtype == torch.bfloat16 and torch.equal(x16, x.to(torch.bfloat16))          # torch rounds to nearest even too
    a16 = ops.cast_bf16(xa)
    b16 = ops.cast_bf16(xb) if xb is not None else None
    kw = dict(scale=(torch.rand(cout, generator=g) + 0.5).to(device), shift=torch.randn(cout, generator=g).to(device),
              residual=torch.randn(reps * m_out, cout, generator=g).to(device), relu=True, replicas=reps)
    ref = ops.spconv_fwd_bf16(xa, w, nbr, m_out, in_b=xb, **kw)
    for kern in ("two_stage", "ring"):                 # the two pair-list kernels for bf16 rows
        got = ops.spconv_fwd_bf16(a16, w, nbr, m_out, in_b=b16, kernel=kern, **kw)
        assert torch.equal(got, ref), (kern, (got - ref).abs().max().item())
    # the wide register-tile kernel sums ONE fp32 chain per output over all offsets and channels (the others: per offset first)
    wide = ops.spconv_fwd_bf16(a16, w, nbr, m_out, in_b=b16, kernel="wide", **kw)
    assert torch.allclose(wide, ref, rtol=1e-4, atol=1e-4), (wide - ref).abs().max().item()
    assert torch.equal(wide, ops.spconv_fwd_bf16(a16, w, nbr, m_out, in_b=b16, kernel="wide", **kw))
    auto = ops.spconv_fwd_bf16(a16, w, nbr, m_out, in_b=b16, **kw)          # the default: wide tiles except on the stride-2 maps
    assert torch.equal(auto, ref if kind == "down" else wide)
    if not split and kind != "down":
        # the input gradient's form: the transposed kernel over the (here: the same, flipped) map; and the weight gradient
        gr = torch.randn(m_out, cout, generator=g).to(device)
        flip = kind == "k3"
        ref_x = ops.spconv_fwd_bf16(gr, w, nbr, m_in, transposed=True, flip=flip)
        for kern in ("two_stage", "ring"):
            assert torch.equal(ops.spconv_fwd_bf16(ops.cast_bf16(gr), w, nbr, m_in, transposed=True, flip=flip, kernel=kern), ref_x)
        x1 = x[:m_in].contiguous()
        ref_w = ops.spconv_bwd_w(x1, gr, nbr, k, bf16=True)
        got_w = ops.spconv_bwd_w(ops.cast_bf16(x1), ops.cast_bf16(gr), nbr, k, bf16=True)
        assert torch.equal(got_w, ref_w), (got_w - ref_w).abs().max().item()


@pytest.mark.parametrize("n_points", [1, 17, 300, 2000])
def test_spconv_bf16_row_kernels_on_tiny_and_ragged_maps(device, n_points):
    """The three kernels for bf16 rows on maps far smaller than a tile, with ragged last tiles, with kernel offsets that have no
    pair at all (a cloud spread thin: most of the 27 offsets are empty) and with one replica: the pair-list kernels bit-identical to
    the fp32-row form, the wide register-tile kernel within 1e-4, and all of them against the float64 oracle on the bf16-rounded
    operands at the fp32 kernel's bar."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(100 + n_points)
    uniq, _, _ = me.voxelize(random_cloud(n_points, 9, 200 + n_points))
    nbr_np = me.kernel_map(uniq, uniq, 3, 1)
    m = uniq.shape[0]
    nbr = dev_i32(nbr_np, device)
    for cin, cout in ((64, 128), (96, 96), (32, 32), (256, 256)):
        x = torch.randn(m, cin, generator=g)
        w = torch.randn(27, cin, cout, generator=g) / np.sqrt(cin * 3)
        xd, wd = x.to(device), w.to(device)
        ref = ops.spconv_fwd_bf16(xd, wd, nbr, m)
        x16 = ops.cast_bf16(xd)
        for kern in ("two_stage", "ring"):
            assert torch.equal(ops.spconv_fwd_bf16(x16, wd, nbr, m, kernel=kern), ref), (kern, cin, cout, m)
        wide = ops.spconv_fwd_bf16(x16, wd, nbr, m, kernel="wide")
        assert torch.allclose(wide, ref, rtol=1e-4, atol=1e-4), (cin, cout, m, (wide - ref).abs().max().item())
        want = me.conv_forward(x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), nbr_np)
        assert torch.allclose(wide.cpu().double(), want, rtol=RTOL, atol=ATOL), (cin, cout, m)
