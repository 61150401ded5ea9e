"""The denoising step WITHOUT a host read (DiffCompletion.read_free): coordinate maps handed
over at their bound with the row counts on the device, every kernel of the fused plan taking them from there, kernel choices
from the previous pyramid's sizes, the sizes published by the device into pinned memory (ops.SizeFeed).

  * kernel level: every operator that gained a device-side row count gives, on the valid rows, BIT FOR BIT what the exact-size
    call gives (tile / row / thin / pair-list convolutions with replicas, epilogue and fused cat; centre + tail over a bounded
    tail map; the kernel-map builders; the conditioning multiply);
  * pyramid level: a read-free pyramid holds the same maps as the one-read pyramid, and the feed delivers its sizes;
  * loop level: completion_loop read-free == the exact-size loop making the same kernel choices (LIDIFF_HINT_LAG twin), bit for
    bit, on a small scene and on the 180 000-point scan; a voided loop (tail map above its pair bound / a condition of another
    size) is detected and redone with exact sizes.
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import random_cloud, record_parity
from oracle import me_cpu as me

pytestmark = pytest.mark.gpu


def dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


def _bounded(t, bound, fill):
    """t's rows at the front of a buffer of `bound` rows whose tail holds `fill` (garbage the kernels must never use)."""
    out = torch.full((bound,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


def test_size_feed_delivers_device_words_in_order(device):
    """ops.SizeFeed: records written by the device (lidiff_publish_words) come back in order with their status word, across
    more records than the ring has slots, and interleaved with host-pushed ones; the copy fallback gives the same."""
    from lidiff_amd import ops
    for copy_mode in (False, True):
        feed = ops.SizeFeed(device)
        if copy_mode:
            feed.dev_base = None
        else:
            assert feed.dev_base is not None, "pinned memory is not mapped into the device's address space"
        status = torch.zeros(1, dtype=torch.int32, device=device)
        seqs = []
        for i in range(3 * feed.SLOTS + 1):
            if i % 5 == 4:
                seqs.append((feed.push_host([i, 2 * i, 7]), 0, [i, 2 * i, 7]))
                continue
            words = torch.tensor([i, i * i, 123456 + i, -1], dtype=torch.int32, device=device)
            status.fill_(i % 3)
            seqs.append((feed.publish(words, status.clone()), i % 3, [i, i * i, 123456 + i, -1]))
        assert feed.get() == (seqs[-1][1], seqs[-1][2])
        for seq, st, words in seqs[-feed.SLOTS:]:
            assert feed.get(seq) == (st, words), seq
        assert feed.drain() is None
        feed.reset()
        assert not feed.has_records()
        feed.publish(torch.tensor([5, 6], dtype=torch.int32, device=device), (status * 0 + ops.STATUS_BOUND))
        assert feed.drain() is not None and "bound" in feed.bad
        feed.reset()
        s = feed.publish(torch.tensor([5, 6], dtype=torch.int32, device=device), status * 0)
        feed.expect(s, {1: 7})
        assert "exact" in feed.drain()


@pytest.mark.parametrize("cin,split,cout,kind", [(64, 0, 128, "k3"), (256, 128, 256, "k3"), (32, 0, 32, "k3"), (96, 0, 96, "k3"),
                                                (64, 0, 64, "k3"), (3, 0, 32, "k3"), (128, 96, 96, "k1"), (32, 0, 64, "k1"),
                                                (256, 0, 256, "k1")])
def test_convolutions_with_the_row_count_on_the_device(device, cin, split, cout, kind):
    """lidiff_spconv_fwd(d_m_out): tables / features / residuals at a BOUND (garbage behind the valid rows), the count on the
    device -- valid output rows bit-identical to the exact-size launch, rows behind the count untouched; tile, row and thin
    kernels, dense and hinted maps, replicas 1 / 2, epilogue, fused cat."""
    from lidiff_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    for cloud, bound_extra in ((random_cloud(3000, 9, 4, batch=2), 700), (random_cloud(900, 40, 5, dup=0.0), 5000)):
        uniq, _, _ = me.voxelize(cloud)
        m = uniq.shape[0]
        bound = m + bound_extra
        nbr_np = me.kernel_map(uniq, uniq, 3, 1) if kind == "k3" else None
        k = 27 if kind == "k3" else 1
        w = (torch.randn(k, cin, cout, generator=g) / np.sqrt(cin * max(1, k // 3))).to(device)
        sc, sh = (torch.rand(cout, generator=g) + 0.5).to(device), torch.randn(cout, generator=g).to(device)
        d_rows = torch.tensor([m], dtype=torch.int32, device=device)
        for reps in (1, 2):
            x = torch.randn(reps, m, cin, generator=g).to(device)
            res = torch.randn(reps, m, cout, generator=g).to(device)
            xb = torch.stack([_bounded(x[r], bound, float("nan")) for r in range(reps)]).reshape(reps * bound, cin)
            rb = torch.stack([_bounded(res[r], bound, float("nan")) for r in range(reps)]).reshape(reps * bound, cout)
            nbr = None if nbr_np is None else dev_i32(nbr_np, device)
            nbr_b = None
            if nbr is not None:
                nbr_b = torch.full((k, bound), -1, dtype=torch.int32, device=device)
                nbr_b[:, :m] = nbr
            for hint in ((False, True) if kind == "k3" and cin % 32 == 0 else (False,)):
                cut = lambda t: (t[:, :split].contiguous(), t[:, split:].contiguous()) if split else (t, None)
                a, b = cut(x.reshape(reps * m, cin))
                ab, bb = cut(xb)
                kw = dict(scale=sc, shift=sh, relu=True, sparse_map=hint, replicas=reps)
                want = ops.spconv_fwd(a, w if k > 1 else w[0], nbr, m, in_b=b, residual=res.reshape(reps * m, cout), **kw)
                got = ops.spconv_fwd(ab, w if k > 1 else w[0], nbr_b, bound, in_b=bb, residual=rb, d_rows=d_rows, rows_hint=m, **kw)
                got = got.reshape(reps, bound, cout)
                assert torch.equal(got[:, :m].reshape(reps * m, cout), want), (cin, cout, kind, reps, hint)
                assert torch.isfinite(got[:, :m]).all()


def test_read_free_pyramid_holds_the_maps_of_the_one_read_pyramid(device):
    """A CoordinateManager with a SizeFeed and read_free: the second pyramid of the role is built without a host read -- its
    rows, inverse map, parents, kernel maps (self, down, up, the up-conv pair lists), tail maps and match indices equal the
    exact pyramid's on the valid rows / columns, the columns behind the counts hold no pair, and the feed delivers the sizes."""
    import lidiff_amd.MinkowskiEngine as ME
    from lidiff_amd import ops
    feed = ops.SizeFeed(device)
    clouds = [random_cloud(20000, 60, 3, batch=1, dup=0.2), random_cloud(20000, 50, 4, batch=1, dup=0.2),
              random_cloud(20000, 8, 5, batch=1, dup=0.1)]
    part = dev_i32(me.floor_to_stride(random_cloud(300, 30, 9), 16), device)
    part = ops.vox_unique(part, torch.zeros(1, dtype=torch.int32, device=device))[0]

    def manager(cloud, free):
        mgr = ME.CoordinateManager(torch.device(device))
        mgr.pyramid = True
        mgr.feed, mgr.read_free = feed, free
        inverse, _ = mgr.insert(dev_i32(cloud, device))
        return mgr, inverse

    exact0, _ = manager(clouds[0], False)                    # the role's first pyramid: with its read, pushes its sizes
    assert exact0.maps[1].count is None and feed.has_records()
    for cloud in clouds[1:]:
        ref_mgr = ME.CoordinateManager(torch.device(device))
        ref_mgr.pyramid = True
        ref_inv, _ = ref_mgr.insert(dev_i32(cloud, device))
        mgr, inv = manager(cloud, True)
        assert mgr.maps[1].count is not None, "the second pyramid of a role must be read-free"
        assert torch.equal(inv, ref_inv)
        n = cloud.shape[0]
        sizes = feed.get()[1]
        overflowed = False
        for lv in range(5):
            ts = 1 << lv
            m = ref_mgr.maps[ts].coords.shape[0]
            assert sizes[lv] == m and int(mgr.count(ts).item()) == m
            assert mgr.maps[ts].coords.shape[0] == n
            assert torch.equal(mgr.maps[ts].coords[:m], ref_mgr.maps[ts].coords)
            if lv:
                assert torch.equal(mgr.parents[ts][:ref_mgr.parents[ts].shape[0]], ref_mgr.parents[ts])
            a, b = mgr.kernel_map(ts, ts, 3), ref_mgr.kernel_map(ts, ts, 3)
            assert a.shape == (27, n) and torch.equal(a[:, :m], b) and bool((a[:, m:] == -1).all())
            if lv < 4:
                mc = ref_mgr.maps[2 * ts].coords.shape[0]
                a, b = mgr.kernel_map(ts, 2 * ts, 2), ref_mgr.kernel_map(ts, 2 * ts, 2)
                assert torch.equal(a[:, :mc], b) and bool((a[:, mc:] == -1).all())
                a, b = mgr.kernel_map(2 * ts, ts, 2, True), ref_mgr.kernel_map(2 * ts, ts, 2, True)
                assert torch.equal(a[:, :m], b) and bool((a[:, m:] == -1).all())
                mgr.UP_ORDER_MIN_ROWS = ref_mgr.UP_ORDER_MIN_ROWS = 16
                got, want = mgr.up_pairs(2 * ts, ts), ref_mgr.up_pairs(2 * ts, ts)
                if want is not None:
                    assert got is not None and torch.equal(got[2], want[2])
                    assert torch.equal(got[0][:m], want[0]) and torch.equal(got[1][:m], want[1])
                    assert torch.equal(torch.sort(got[1]).values.cpu(), torch.arange(n, dtype=torch.int32))      # a permutation
            # tail maps: pair list and CSR of the valid rows
            tm, tr = mgr.tail_map(ts), ref_mgr.tail_map(ts)
            assert tm.bounded and torch.equal(tm.off, tr.off)
            if tr.n > tm.n:            # more pairs than the bound (a DENSE map: centre + tail is never chosen for it): flagged, not overrun
                overflowed = True
                assert int(mgr.status.item()) & ops.STATUS_BOUND
            elif tr.n:
                assert torch.equal(tm.pair_in[:tr.n], tr.pair_in) and torch.equal(tm.idx[:tr.n], tr.idx)
                assert torch.equal(tm.ptr[:m + 1], tr.ptr)
            # the part -> full match with the count on the device
            got = ops.nn_match_dev(mgr.maps[ts].coords, mgr.count(ts), part)
            assert torch.equal(got[:m], ops.nn_match(ref_mgr.maps[ts].coords, part))
        if overflowed:
            with pytest.raises(RuntimeError, match="bound"):
                mgr.check()
        else:
            mgr.check()
    assert overflowed, "the dense cloud was meant to exceed the tail-pair bound"


def _pipe(device, models, steps, **flags):
    from lidiff_amd.pipeline import DiffCompletion
    enc, unet, refine, _ = models
    pipe = DiffCompletion(denoising_steps=steps, cond_weight=6.0, device=device)
    pipe.partial_enc, pipe.model, pipe.model_refine = enc, unet, refine
    for k, v in flags.items():
        setattr(pipe, k, v)
    pipe.new_scheduler()
    return pipe


def _loop(pipe, scan, x0, zs):
    out = pipe.completion_loop(scan, pipe.points_to_tensor(x0, role="x_t"), pipe.points_to_tensor(scan, role="cond"),
                               pipe.points_to_tensor(torch.zeros_like(scan), role="uncond"), noises=zs)
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def models(device):
    from conftest import build_seeded_models
    enc, unet, refine = build_seeded_models(42)
    return enc.to(device).eval(), unet.to(device).eval(), refine.to(device).eval(), None


@pytest.mark.parametrize("n_base,steps,sigma", [(1500, 6, 0.6), (18000, 4, 1.0), (18000, 3, 0.08)])
def test_read_free_loop_equals_the_exact_loop_bit_for_bit(device, models, fps_scan, n_base, steps, sigma):
    """DiffCompletion.completion_loop without host reads (from every role's second pyramid on) against the exact-size loop that
    makes the SAME kernel choices (hint_lag: the previous pyramid's sizes decide): the same points, BIT FOR BIT, on a small
    scene and on the 180 000-point scan at the dense and at the sparse end of the trajectory; run twice (stream hazards);
    and against the round-4 loop (each step's own sizes decide) within fp32 summation-order noise."""
    scan = torch.from_numpy(np.tile(fps_scan[:n_base], (10, 1))).double()[None].to(device)
    g = torch.Generator(device="cpu").manual_seed(11)
    x0 = (scan.cpu() + sigma * torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device)
    zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(device) for _ in range(steps)]
    twin = _loop(_pipe(device, models, steps, read_free=False, hint_lag=True), scan, x0, zs)
    assert np.isfinite(twin).all()
    for _ in range(2):
        pipe = _pipe(device, models, steps, read_free=True)
        with warnings.catch_warnings():
            warnings.simplefilter("error")              # a voided loop would be redone behind a warning: that is a failure here
            free = _loop(pipe, scan, x0, zs)
        assert pipe._feeds["x_t"].seq - pipe._feeds["x_t"].seq_base == steps, "x_t's pyramids did not go through the feed"
        d = np.abs(free - twin).max(axis=1)
        assert np.array_equal(free, twin), (np.count_nonzero(d), d.size, d.max())
    own = _loop(_pipe(device, models, steps, read_free=False, hint_lag=False), scan, x0, zs)
    err = np.abs(free - own).max(axis=1)
    record_parity(f"read_free_vs_round4_loop_{10 * n_base}pts_T{steps}_sigma{sigma}", max_err_m=float(err.max()),
                  share_above_1e_5=float(np.mean(err > 1e-5)))
    assert np.median(err) <= 1e-5, np.median(err)


def test_a_voided_read_free_loop_is_redone_with_exact_sizes(device, models, fps_scan, monkeypatch):
    """What the host assumes in a read-free step is checked when the device's sizes arrive: (1) a tail map above its pair bound
    (forced: a bound of 1 % of the rows) raises STATUS_BOUND -- completion_loop warns and redoes the loop with exact sizes: the
    result is the exact loop's; (2) a condition whose latent has another size than the step before (a different scan handed to
    denoise_step mid-loop) is reported by read_free_check()."""
    from lidiff_amd import ops
    steps = 4
    scan = torch.from_numpy(np.tile(fps_scan[:2000], (10, 1))).double()[None].to(device)
    g = torch.Generator(device="cpu").manual_seed(12)
    x0 = (scan.cpu() + torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device)
    zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(device) for _ in range(steps)]
    want = _loop(_pipe(device, models, steps, read_free=False), scan, x0, zs)
    monkeypatch.setattr(ops, "TAIL_PAIR_BOUND", 0.01)
    pipe = _pipe(device, models, steps, read_free=True)
    with pytest.warns(UserWarning, match="voided"):
        got = _loop(pipe, scan, x0, zs)
    assert np.array_equal(got, want)
    monkeypatch.setattr(ops, "TAIL_PAIR_BOUND", 3)
    # (2) another condition in the middle of a loop
    pipe = _pipe(device, models, steps, read_free=True)
    pipe.read_free_reset()
    other = torch.from_numpy(np.tile(fps_scan[3000:4500], (10, 1))).double()[None].to(device)[:, :scan.shape[1]]
    x_t = pipe.points_to_tensor(x0, role="x_t")
    x_c, x_u = pipe.points_to_tensor(scan, role="cond"), pipe.points_to_tensor(torch.zeros_like(scan), role="uncond")
    ts = pipe.dpm_scheduler.host_timesteps
    for i in range(3):
        x_t, x_c, x_u = pipe.denoise_step(scan, x_t, x_c, x_u, ts[i], zs[i], next_t=ts[i + 1])
        if i == 0:
            assert pipe.read_free_check() is None
        if i == 1:                                     # swap the condition: its sizes no longer are the previous step's
            x_c = pipe.points_to_tensor(other, role="cond")
    why = pipe.read_free_check()
    assert why is not None and "cond" in why and "exact" in why, why


def test_two_piece_fp16_scan_is_redone_on_three_pieces_when_a_value_leaves_its_range(device, models, fps_scan, monkeypatch):
    """The opt-in two-piece fp16 mode (ops.split_pieces(2)) reports a value beyond fp16's range through a device flag; completion_loop
    then warns and redoes the scan -- same inputs, scheduler state and draws -- on the default three bf16 pieces (every split-operand
    launch after the warning runs on three pieces; the result is the default mode's to the bars of the loop tests -- not bit for
    bit: the redone loop reuses the input fields' maps, so its second step takes exact sizes where a fresh loop takes hints).  The
    flag is raised by hand here: the networks' activations stay far inside the range.  Without the flag the mode runs to the end
    and agrees with the default to the same bars."""
    from lidiff_amd import ops
    steps = 3
    scan = torch.from_numpy(np.tile(fps_scan, (10, 1))).double()[None].to(device)
    g = torch.Generator(device="cpu").manual_seed(21)
    x0 = (scan.cpu() + torch.randn(scan.shape, generator=g, dtype=torch.float64)).to(device)
    zs = [torch.randn(scan.shape, generator=g, dtype=torch.float64).to(device) for _ in range(steps)]
    with ops.split_pieces(3):
        want = _loop(_pipe(device, models, steps, read_free=True), scan, x0, zs)
    used = []
    inner = ops.spconv_fwd_split3
    monkeypatch.setattr(ops, "spconv_fwd_split3", lambda *a, **k: (used.append(ops.SPLIT_PIECES), inner(*a, **k))[1])
    with ops.split_pieces(2):
        fast = _loop(_pipe(device, models, steps, read_free=True), scan, x0, zs)
        n = len(used)
        assert n and set(used) == {2}
        ops.split_status(device).fill_(ops.STATUS_F16_RANGE)
        with pytest.warns(UserWarning, match="three bf16 pieces"):
            redone = _loop(_pipe(device, models, steps, read_free=True), scan, x0, zs)
        assert set(used[n:2 * n]) == {2} and len(used) > 2 * n and set(used[2 * n:]) == {3}       # the attempt on 2 pieces, the redo on 3
    for name, got in (("redone", redone), ("two_piece", fast)):
        err = np.abs(got - want).max(axis=1)
        record_parity(f"{name}_fp16_loop_vs_default_T{steps}", max_err_m=float(err.max()), median_err_m=float(np.median(err)))
        assert np.median(err) <= 1e-5, (name, np.median(err))
