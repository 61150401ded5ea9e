"""Scan-completion inference pipeline: mirror of
/root/reference/lidiff/tools/diff_completion_pipeline.py (``DiffCompletion``, lines 15-169;
``load_pcd`` 171-177) without the Lightning / open3d / diffusers dependencies.

Same method names, argument meaning and numerical behaviour:
  points_to_tensor 68-84 (divides the batch column by the resolution too, App. D.2),
  reset_partial_pcd 86-90, preprocess_scan 92-105, postprocess_scan 107-115,
  complete_scan 117-132, refine_forward 134-138, forward 140-146,
  classfree_forward 148-153, completion_loop 155-169.
Host-side deviations (no effect on results): no ``torch.cuda.empty_cache()`` calls (App. D.10),
timesteps are walked as host integers (no per-step device sync inside the scheduler), and a
fresh scheduler state per scan (App. D.4).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import MinkowskiEngine as ME
from . import minkunet as minknet
from .schedulers import DPMSolverMultistepScheduler

DEFAULT_HPARAMS = {   # lidiff/config/config.yaml
    "data": {"resolution": 0.05, "num_points": 180000, "max_range": 50.0},
    "train": {"uncond_prob": 0.1, "uncond_w": 6.0, "lr": 1e-4, "batch_size": 2},
    "diff": {"beta_start": 3.5e-5, "beta_end": 0.007, "beta_func": "linear", "t_steps": 1000,
             "s_steps": 50, "reg_weight": 5.0},
    "model": {"out_dim": 96},
}


class DiffCompletion(nn.Module):
    def __init__(self, diff_path=None, refine_path=None, denoising_steps=50, cond_weight=6.0,
                 hparams=None, device="cuda"):
        super().__init__()
        ckpt_diff = torch.load(diff_path, map_location="cpu") if diff_path else None
        self.hparams = _merge(DEFAULT_HPARAMS, (ckpt_diff or {}).get("hyper_parameters", hparams or {}))
        assert denoising_steps <= self.hparams["diff"]["t_steps"], \
            f"The number of denoising steps cannot be bigger than T={self.hparams['diff']['t_steps']}"
        out_dim = self.hparams["model"]["out_dim"]
        self.partial_enc = minknet.MinkGlobalEnc(in_channels=3, out_channels=out_dim)
        self.model = minknet.MinkUNetDiff(in_channels=3, out_channels=out_dim)
        self.model_refine = minknet.MinkUNet(in_channels=3, out_channels=3 * 6)
        if ckpt_diff is not None:
            self.load_state_dict(ckpt_diff["state_dict"], strict=False)
        if refine_path:
            self.load_state_dict(torch.load(refine_path, map_location="cpu")["state_dict"], strict=False)
        self.eval()
        self.to(device)
        self.device = torch.device(device)

        self.hparams["diff"]["s_steps"] = denoising_steps
        self.hparams["train"]["uncond_w"] = cond_weight
        self.hparams["data"]["max_range"] = 50.0
        self.w_uncond = cond_weight
        self.pair_cfg = True       # run the CFG pair as one stacked pass (False: two forwards, as the reference does)
        # x_cond / x_uncond are rebuilt from the same points every step (pipeline:86-90,166), so
        # partial_enc(x_cond) and partial_enc(x_uncond) are step-invariant; with this flag completion_loop encodes
        # them once per scan (bit-identical results: eval mode, deterministic kernels).  Needs pair_cfg.
        self.cache_condition = False
        self.new_scheduler()

    def new_scheduler(self):
        d = self.hparams["diff"]
        self.dpm_scheduler = DPMSolverMultistepScheduler(
            num_train_timesteps=d["t_steps"], beta_start=d["beta_start"], beta_end=d["beta_end"],
            beta_schedule="linear", algorithm_type="sde-dpmsolver++", solver_order=2)
        self.dpm_scheduler.set_timesteps(d["s_steps"])
        self.dpm_scheduler.to(self.device)

    # The boundary between two steps -- guidance, DPM-Solver++ update, the next field's points and voxel coordinates -- as ONE
    # launch (ops.cfg_dpm_step / step.hip) instead of ~25 elementwise torch launches, and points_to_tensor as one launch
    # (ops.points_to_field) instead of six; same values bit for bit (test_fused_step_boundary_equals_the_torch_sequence).
    # fused_step = False: the torch sequence.
    fused_step = True

    # -- a denoising step WITHOUT a host read ---------------------------------------------
    # Every step rebuilds three coordinate pyramids, and the sizes of their maps used to come back to the host (one blocking read
    # per pyramid) because they shaped every later launch and allocation: the main queue idled ~1 ms per step behind x_t's read and
    # the host never got ahead of the device.  With read_free the fields of a loop carry a ROLE ("x_t" / "cond" / "uncond"); the
    # first pyramid of a role is built with its read, every later one without: maps are handed over at their bound (the point
    # count) with the row counts on the device -- every kernel of the fused plan takes them from there --, kernel choices are
    # made from the sizes of the role's PREVIOUS pyramid (deterministic: always the previous one), and the device publishes the
    # sizes into pinned memory (ops.SizeFeed) where the host finds them one step later.  What the host assumed (tail-pair bounds,
    # the condition latent's row count) is checked then; a failed check voids the loop, which is redone with exact sizes
    # (completion_loop does that itself; callers of denoise_step ask read_free_check()).  Results are independent of the mode
    # up to the kernel choices; LIDIFF_HINT_LAG=1 runs the EXACT-size path with the same choices (the bit-for-bit twin).
    read_free = os.environ.get("LIDIFF_READ_FREE", "1") != "0"
    hint_lag = os.environ.get("LIDIFF_HINT_LAG", "0") == "1"

    def _feed(self, role):
        from . import ops
        feeds = self.__dict__.setdefault("_feeds", {})
        if role not in feeds:
            feeds[role] = ops.SizeFeed(self.device)
        return feeds[role]

    def _loop_status(self):
        """ONE status word for all the coordinate managers of a loop's fields (roles): a bound exceeded by a kernel that runs long
        after its pyramid was published (a tail map filled when a layer first asks for it) still reaches the host -- with the
        next pyramid's record, and through read_free_check() at the latest."""
        if self.__dict__.get("_status") is None:
            self._status = torch.zeros(1, dtype=torch.int32, device=self.device)
        return self._status

    def read_free_reset(self):
        """A new scan: the first pyramid of every role is built with a host read again."""
        for f in self.__dict__.get("_feeds", {}).values():
            f.reset()
        if self.__dict__.get("_status") is not None:
            # the word is shared by every field of the loop, the ones voxelised BEFORE it included: what they flagged (a
            # coordinate outside the key range, a full table) is raised here, not erased; one read per scan
            from . import ops
            s = int(self._status.item())
            self._status.zero_()
            if s & ops.STATUS_KEY_RANGE:
                raise RuntimeError("coordinate outside [-32768, 32767]: not representable in the 64-bit key")
            if s & ops.STATUS_HASH_FULL:
                raise RuntimeError("coordinate hash table overflow")

    def read_free_check(self):
        """None, or why the host-read-free steps since the last reset are void (waits for the sizes the device still owes and
        reads the loop's status word: one synchronisation, at the END of a loop)."""
        from . import ops
        for role, f in self.__dict__.get("_feeds", {}).items():
            bad = f.drain()
            if bad is not None:
                return f"{role}: {bad}"
        if self.__dict__.get("_status") is not None and any(f.has_records() for f in self._feeds.values()):
            if int(self._status.item()) & ops.STATUS_BOUND:
                return "a device-side count exceeded its bound (the loop's status word)"
        return None

    # pipeline:68-84
    def points_to_tensor(self, points, role=None):
        """role: "x_t" / "cond" / "uncond" inside a denoising loop (see read_free above); None: a field on its own."""
        if (self.fused_step and isinstance(points, torch.Tensor) and points.is_cuda and points.dim() == 3
                and points.shape[2] == 3 and points.dtype in (torch.float32, torch.float64)):
            from . import ops
            feats, coords = ops.points_to_field(points.detach(), self.hparams["data"]["resolution"], scale_batch_column=True)
            return self._make_field(feats, coords, role)
        x_feats = ME.utils.batched_coordinates(list(points[:]), dtype=torch.float32, device=self.device)
        x_coord = torch.round(x_feats / self.hparams["data"]["resolution"])
        return self._make_field(x_feats[:, 1:], x_coord, role)

    def _make_field(self, feats, coords, role=None):
        field = ME.TensorField(features=feats, coordinates=coords,
                               quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE,
                               minkowski_algorithm=ME.MinkowskiAlgorithm.SPEED_OPTIMIZED, device=self.device)
        mgr = field.coordinate_manager
        mgr.pyramid = self.single_read and field.F.device.type == "cuda"
        if (role is not None and mgr.pyramid and (self.read_free or self.hint_lag) and self.pair_cfg and minknet._FUSION
                and not self.training and feats.shape[0] >= 1):
            mgr.feed = self._feed(role)
            mgr.status = self._loop_status()
            mgr.read_free = bool(self.read_free)
            mgr.hint_lag = bool(self.hint_lag) and not self.read_free
            if role != "x_t":            # the condition's latent shapes host-side work (MLP tables, match targets): exact rows
                mgr.exact_rows = (ME.CoordinateManager.MAX_STRIDE,)
        if self.overlap_maps and field.F.device.type == "cuda":
            field.ready = torch.cuda.Event()         # its points exist once the current stream gets here
            field.ready.record(torch.cuda.current_stream(self.device))
        return field

    # -- overlap of the coordinate pipeline with the convolutions ---------------------------------------------------
    # Voxelising a field and building its maps is a chain of small, latency-bound kernels with a host read of every map
    # size; on the stream of the convolutions each read drains the queue.  With overlap_maps the chain of a field runs
    # on a side stream while the main stream is busy with another tensor's convolutions: the conditions of step i + 1
    # under the UNet of step i, the maps of x_t under the condition encoders.  Every step still builds everything anew.
    # x_t's maps built ON DEMAND on the side streams while the network's first layers already run (round 3): with the condition
    # encoders issued one step ahead nothing else hides the chain behind x_t's points (voxelise -> 4 strided maps -> 13 kernel
    # maps -> tail maps -> up orders -> 5 matches: 2.2-3 ms with its map-size reads, profiles/r03_step_boundary_trace.txt),
    # but the stem needs only the first ~0.4 ms of it.  lazy_x_t = False: the whole pyramid first (round-2 behaviour).
    lazy_x_t = True
    # every field's pyramid (voxel map, four strided maps, the first two levels' kernel_size-3 maps and tail-map counts) queued
    # with the row counts staying on the device and ONE host read at its end (ops.build_pyramid) instead of seven
    single_read = True
    overlap_maps = True
    eager_maps = True              # False: maps are built when a layer first asks

    # (Round 4 measured the main path on a HIGH-priority stream, so that the side streams' kernels would only fill the chip's idle
    # corners instead of taking compute units from the convolutions: 37.86 vs 38.01 ms per step with per-launch events, 37.84 vs
    # 37.78 without -- no effect, not kept.)
    def _streams(self):
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=self.device)
            # second side stream: the part -> full matches next to the kernel maps.  (Round 5 measured this stream with a CU mask --
            # hipExtStreamCreateWithCUMask, every 8th / 4th / 2nd compute unit left to the other queues, so that the chip-filling
            # match kernels would not hold back the small kernels of the map-building chain: 41.0 vs 37.5 ms per step, on the
            # default and on an explicit main stream alike.  A masked queue costs far more than it frees; not kept.)
            self._side2 = torch.cuda.Stream(device=self.device)
            self._side3 = torch.cuda.Stream(device=self.device)     # third: the deeper levels of a host-read-free pyramid
        return torch.cuda.current_stream(self.device), self._side

    def prepare(self, field, tail_maps=True, also=None, up_orders=False):
        """Voxelise `field` and build ALL its maps now -- on the side stream with overlap_maps, else on the current one
        (no-op if already done).  Building eagerly also gives every level its sparse-map hint: asked lazily, a level's hint
        needs the next coarser map, which the encoder half of a network has not built yet."""
        if (not self.eager_maps or field.prepared is not None or field.F.device.type != "cuda"
                or getattr(field, "_keep", None) is not None):
            return field
        if not self.overlap_maps:                    # same work, same maps (and the same kernel choices), on this stream
            with torch.no_grad():
                field._keep = field.sparse()
                field.coordinate_manager.prebuild(tail_maps=tail_maps, up_orders=up_orders)
                if also is not None:
                    for ts in sorted(field.coordinate_manager.maps):
                        also(field, ts)
            return field
        main, side = self._streams()
        if field.ready is None:                      # produced on the main stream just now
            field.ready = torch.cuda.Event()
            field.ready.record(main)
        side.wait_event(field.ready)
        with torch.cuda.stream(side), torch.no_grad():
            self._stamp("prepare: enter")
            sp = field.sparse()
            self._stamp("prepare: voxelised")
            joined = None
            if also is not None:
                # `also(field, ts)` (the part -> full match of one level: a brute-force search, five of them ~1.1 ms) needs
                # coordinates only: each level's search is queued on a second side stream as soon as that level's map exists,
                # NEXT TO the rest of the strided maps, the kernel maps, tail maps and up orders of this stream.  The chain behind
                # x_t's points is what the network waits for once the condition encoders are through (measured in round 3:
                # 2.45 ms in series on an idle GPU, of which the searches are 1.14).
                mgr, ts = field.coordinate_manager, 1
                while True:
                    forked = torch.cuda.Event()
                    forked.record(side)
                    self._side2.wait_event(forked)
                    with torch.cuda.stream(self._side2):
                        also(field, ts)
                    if ts == 16:
                        break
                    ts = mgr.stride(ts, 2)
                joined = torch.cuda.Event()
                joined.record(self._side2)
                self._stamp("prepare: strides + matches queued")
            field.coordinate_manager.prebuild(tail_maps=tail_maps, up_orders=up_orders)
            self._stamp("prepare: maps built")
            if joined is not None:
                side.wait_event(joined)
            field.prepared = torch.cuda.Event()
            field.prepared.record(side)
        field._keep = sp
        return field

    def _match_level(self, field, parts, ts, ahead=False):
        """ahead: on a stream of its own -- the result carries an event, and whoever uses this level's match waits for that
        event only (MinkUNetDiff.match_index), not for the matches of the other levels queued behind it."""
        mgr = field.coordinate_manager
        for part in parts:
            if part.C.shape[0] > 1:                  # a one-voxel part (the unconditional branch) needs no match
                self.model.match_index(ME.SparseTensor(self._empty(), tensor_stride=ts, coordinate_manager=mgr), part,
                                       ahead=ahead)

    # x_t's part -> full matches queued INSIDE its pyramid chain, reading the row counts from the device (lidiff_nn_match_dev):
    # they run while the host is still blocked in the pyramid's size read, and the host goes from that read straight to the first
    # convolutions instead of first queuing five searches (profiles/r04_step_boundary.txt).  match_in_chain = False: behind the read.
    match_in_chain = True

    def _match_level_dev(self, parts, ts, rows_bound, d_count):
        from . import ops
        out = []
        for part in parts:
            if part.C.shape[0] > 1:
                idx = ops.nn_match_dev(rows_bound, d_count, part.C)
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(self.device))
                out.append((("match", ts, id(part.coordinate_manager), part.tensor_stride), part.coordinate_manager, idx, done))
        return out

    def _empty(self):
        if getattr(self, "_empty_t", None) is None:
            self._empty_t = torch.empty((0, 0), device=self.device)
        return self._empty_t

    def _adopt(self, field):
        """Make a prepared field's tensors safe to use on the current stream."""
        if field.prepared is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_event(field.prepared)
            field.coordinate_manager.record_stream(main)
            for t in (field._keep.F, field.inverse_mapping):
                t.record_stream(main)
            field.prepared = None
            field.ready = torch.cuda.Event()         # consumed: a later prepare() of the same field is a no-op anyway
        return field

    # The encoders of the NEXT step's conditions and (with next_t) its conditioning tables, queued on the side stream under this
    # step's UNet as well (round 3).  Nothing is cached: every step still rebuilds its conditions from the points and encodes them
    # (pipeline:86-90,140-146) -- the work is only issued one step ahead, where ~150 launches that cannot fill the chip (3.2 ms on
    # the main stream, profiles/r03_step_timeline.txt) run beside convolutions that can.  Off: encode_ahead = False.
    encode_ahead = True

    # pipeline:86-90
    def reset_partial_pcd(self, x_part, x_uncond, next_t=None):
        if not self.overlap_maps or x_part.F.device.type != "cuda":
            x_part = self.points_to_tensor(x_part.F.reshape(1, -1, 3).detach(), role="cond")
            x_uncond = self.points_to_tensor(torch.zeros_like(x_part.F.reshape(1, -1, 3)), role="uncond")
            return self.prepare(x_part), self.prepare(x_uncond, tail_maps=False)
        # The next step's conditions are rebuilt from the same points (pipeline:86-90), which were there before this step's
        # network was queued: the whole rebuild -- batched coordinates, rounding, voxelisation, maps -- runs on the side
        # stream, ordered after the START of this step (not after its UNet), i.e. under the UNet.
        main, side = self._streams()
        start = getattr(self, "_step_start", None)
        if start is None:
            start = torch.cuda.Event()
            start.record(main)
        side.wait_event(start)
        # the old field's points were allocated on the main stream and are dropped when this step returns: tell the caching
        # allocator that the side stream reads them, or the block could be handed out again while the side stream still reads
        x_part.F.record_stream(side)
        with torch.cuda.stream(side):
            pts = x_part.F.reshape(1, -1, 3).detach()
            x_part = self.points_to_tensor(pts, role="cond")
            x_uncond = self.points_to_tensor(torch.zeros_like(pts), role="uncond")
            self.prepare(x_part)
            self.prepare(x_uncond, tail_maps=False)       # one voxel: nothing to gain from tail maps
            if (self.encode_ahead and self.pair_cfg and not self.cache_condition
                    and getattr(self, "_warm_state", None) == self._encoder_state() and not self.partial_enc.training):
                # (only while the encoder's weights are the ones a main-stream pass has packed / folded: _encoder_state)
                with torch.no_grad():
                    parts = (self.partial_enc(x_part), self.partial_enc(x_uncond))
                    cond = None
                    if next_t is not None:
                        t = torch.full((1,), int(next_t), dtype=torch.int64, device=self.device)
                        cond = self.model.precompute_conditioning(parts, t)
                    done = torch.cuda.Event()
                    done.record(side)
                x_part._encoded = (parts, cond, None if next_t is None else int(next_t), done, x_uncond)
        return x_part, x_uncond

    # pipeline:92-105
    def preprocess_scan(self, scan):
        scan = np.asarray(scan)
        dist = np.sqrt(np.sum(scan ** 2, -1))
        scan = scan[(dist < self.hparams["data"]["max_range"]) & (dist > 3.5)][:, :3]
        pts = torch.tensor(scan, device=self.device)                     # float64 like the reference
        keep = farthest_point_sample(pts, int(self.hparams["data"]["num_points"] / 10))
        scan = pts[keep].repeat(10, 1)
        return scan[None, :, :]

    # pipeline:107-115
    def postprocess_scan(self, completed_scan, input_scan):
        dist = np.sqrt(np.sum(completed_scan ** 2, -1))
        post_scan = completed_scan[dist < self.hparams["data"]["max_range"]]
        max_z = input_scan[..., 2].max().item()
        min_z = (input_scan[..., 2].mean() - 2 * input_scan[..., 2].std()).item()
        return post_scan[(post_scan[:, 2] < max_z) & (post_scan[:, 2] > min_z)]

    # pipeline:117-132
    def complete_scan(self, scan, generator=None, timings: dict | None = None):
        """timings: a dict that receives the seconds spent in preprocess (range filter + FPS), the denoising loop, and
        post-filter + refinement (each ended by a device synchronisation; None = no extra synchronisation)."""
        import time

        def lap(key, t0):
            if timings is not None:
                torch.cuda.synchronize(self.device)
                timings[key] = timings.get(key, 0.0) + time.perf_counter() - t0
            return time.perf_counter()
        t0 = time.perf_counter()
        scan = self.preprocess_scan(scan)
        t0 = lap("preprocess_s", t0)
        x_feats = scan + torch.randn(scan.shape, device=self.device, generator=generator, dtype=scan.dtype)
        self.read_free_reset()
        x_full = self.points_to_tensor(x_feats, role="x_t")
        x_cond = self.points_to_tensor(scan, role="cond")
        x_uncond = self.points_to_tensor(torch.zeros_like(scan), role="uncond")
        self.new_scheduler()
        completed_scan = self.completion_loop(scan, x_full, x_cond, x_uncond)
        t0 = lap("denoise_s", t0)
        post_scan = self.postprocess_scan(completed_scan, scan)
        refine_in = self.points_to_tensor(torch.as_tensor(post_scan[None, :, :]))
        offset = self.refine_forward(refine_in).reshape(-1, 6, 3)
        refine_complete_scan = post_scan[:, None, :] + offset.cpu().numpy()
        lap("refine_s", t0)
        return refine_complete_scan.reshape(-1, 3), post_scan

    # pipeline:134-138
    def refine_forward(self, x_in):
        with torch.no_grad():
            return self.model_refine(x_in)

    # pipeline:140-146
    def forward(self, x_full, x_full_sparse, x_part, t):
        with torch.no_grad():
            # a field that reset_partial_pcd / prepare built on the side stream: join before the encoder reads its maps
            part_feat = self.partial_enc(self._adopt(x_part))
            out = self.model(x_full, x_full_sparse, part_feat, t)
        return out.reshape(t.shape[0], -1, 3)

    # pipeline:148-153
    def _encoder_state(self):
        """Identity of partial_enc's tensors as the packed-weight / folded-BatchNorm caches key them (storage + version):
        changes with load_state_dict, an optimizer step, .to(), copy_ -- anything after which the next encoder pass
        re-packs weights and re-folds BatchNorm."""
        ver = ptrs = 0
        for t in list(self.partial_enc.parameters()) + list(self.partial_enc.buffers()):
            ver += t._version
            ptrs ^= t.data_ptr()
        return ver, ptrs, self.partial_enc.training

    def encode_conditions(self, x_cond, x_uncond):
        ahead = getattr(x_cond, "_encoded", None)
        if ahead is not None and ahead[4] is x_uncond:
            # encoded one step ahead on the side stream (reset_partial_pcd): join, hand the tensors over to this stream
            parts, cond, t_next, done, _ = ahead
            x_cond._encoded = None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(done)
            self._adopt(x_cond), self._adopt(x_uncond)
            for q in parts:
                q.F.record_stream(main)
            if cond is not None:
                for v in cond.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(main)
            self._cond_ahead = (cond, t_next, parts)
            return parts
        with torch.no_grad():
            self.prepare(x_cond)                     # no-ops for fields reset_partial_pcd has prepared already
            self.prepare(x_uncond, tail_maps=False)
            state = self._encoder_state()
            if (self.overlap_maps and getattr(self, "_warm_state", None) == state and not state[2]
                    and x_uncond.prepared is not None):
                # the unconditional branch encodes ONE voxel: ~70 launches that cannot fill the chip.  They run on the side
                # stream (where the field's maps were built) while the main stream encodes the condition.  Only while the
                # encoder's tensors are the ones a pass on the MAIN stream has already packed / folded (_warm_state): after
                # any change of the weights the next pass runs on the main stream alone, so that the caches are (re)built
                # in stream order and never first touched from the side stream.
                main, side = self._streams()
                with torch.cuda.stream(side):
                    e_un = self.partial_enc(x_uncond)
                    done = torch.cuda.Event()
                    done.record(side)
                e_c = self.partial_enc(self._adopt(x_cond))
                main.wait_event(done)
                self._adopt(x_uncond)
                e_un.F.record_stream(main)
                return e_c, e_un
            out = self.partial_enc(self._adopt(x_cond)), self.partial_enc(self._adopt(x_uncond))
            self._warm_state = state                 # packed weights / folded BatchNorm now exist, queued on the main stream
            return out

    # tools/step_timeline.py: a list here collects (step start, conditions encoded, x_t adopted, network done) events
    timeline = None
    host_stamps = None          # tools/step_timeline.py --host: (label, time.perf_counter()) of the host thread

    def _stamp(self, label):
        if self.host_stamps is not None:
            import time
            self.host_stamps.append((label, time.perf_counter()))

    def _mark(self, marks):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self.device))
            marks.append(e)

    # (Round 4 measured the conditional and the unconditional forward of a step on TWO streams instead of one stacked pass -- every
    # convolution two launches of half the tiles, meant to fill each other's last round of workgroups (a stacked launch of 5-7 rounds
    # leaves 7-20 % of the chip idle in its last one): bit-identical results, 41.0 vs 38.0 ms per step.  Half-size launches double the
    # latency-bound small kernels and the two queues do not share the chip evenly; the stacked pass stays.)
    def classfree_forward(self, x_t, x_cond, x_uncond, t, parts=None, t_host=None):
        """t_host: the timestep as a host integer when the caller has it (saves nothing but lets the tables that were computed
        one step ahead be matched to this step without reading t back from the device)."""
        e_cond, e_uncond = self.classfree_pair(x_t, x_cond, x_uncond, t, parts, t_host)
        return e_uncond + self.w_uncond * (e_cond - e_uncond)

    def classfree_pair(self, x_t, x_cond, x_uncond, t, parts=None, t_host=None):
        """The two network outputs (conditional, unconditional) [B, N, 3] of classfree_forward, before the guidance mix."""
        marks = [] if self.timeline is not None else None
        self._stamp("step: enter")
        self._mark(marks)
        if self.overlap_maps and x_t.F.device.type == "cuda":
            self._step_start = torch.cuda.Event()
            self._step_start.record(torch.cuda.current_stream(self.device))
        with torch.no_grad():
            if self.pair_cfg:
                # same arithmetic as the two forwards below, but the conditional / unconditional pair shares one
                # pass over x_t's maps: every sparse conv is ONE launch with two stacked feature matrices
                if parts is None:
                    parts = self.encode_conditions(x_cond, x_uncond)      # queued on the main stream ...
                self._mark(marks)
                self._stamp("step: conditions")
                # the conditioning tables of all eight levels need only the latents and t: queued now, while the host still has
                # a lead over the GPU (prepare() below blocks it on the map sizes of x_t)
                ahead, self._cond_ahead = getattr(self, "_cond_ahead", None), None
                if ahead is not None and ahead[0] is not None and ahead[2] is parts and t_host is not None and ahead[1] == int(t_host):
                    cond = ahead[0]                  # computed with the encoders, one step ahead
                else:
                    cond = self.model.precompute_conditioning(parts, t)
                # ... x_t's maps meanwhile, on the side stream -- and the part -> full matches of every level, which need
                # only coordinates (the condition's coarsest map and x_t's maps), not the encoders' features
                lazy = (self.lazy_x_t and self.overlap_maps and self.eager_maps and x_t.F.device.type == "cuda"
                        and x_t.prepared is None and getattr(x_t, "_keep", None) is None and x_t.inverse_mapping is None)
                if lazy:
                    main, side = self._streams()
                    if x_t.ready is None:
                        x_t.ready = torch.cuda.Event()
                        x_t.ready.record(main)
                    x_t.F.record_stream(side), x_t.C.record_stream(side)
                    mgr = x_t.coordinate_manager
                    mgr.lane_up_orders = bool(minknet._UP_ORDERED)
                    mgr.set_async(side, self._side2, ready=x_t.ready, on_level=lambda ts: self._match_level(x_t, parts, ts, ahead=True),
                                  on_level_dev=(lambda ts, rows, cnt: self._match_level_dev(parts, ts, rows, cnt))
                                  if self.match_in_chain else None, side3=self._side3)
                    try:
                        x_t_sparse = x_t.sparse()
                        self._mark(marks)
                        self._stamp("step: x_t voxelised")
                        e_cond, e_uncond = self.model(x_t, x_t_sparse, parts, t, cond=cond)
                    finally:
                        mgr.clear_async()
                    # everything the side streams allocated for this field is read on this stream until the step ends
                    mgr.record_stream(main)
                    for tt in (x_t_sparse.F, x_t.inverse_mapping):
                        tt.record_stream(main)
                else:
                    self.prepare(x_t, also=lambda f, ts: self._match_level(f, parts, ts), up_orders=minknet._UP_ORDERED)
                    self._stamp("step: x_t prepared")
                    x_t_sparse = self._adopt(x_t).sparse()
                    self._mark(marks)
                    self._stamp("step: adopted")
                    e_cond, e_uncond = self.model(x_t, x_t_sparse, parts, t, cond=cond)
                self._mark(marks)
                self._stamp("step: network queued")
                if marks is not None:
                    self.timeline.append(marks)
                return e_cond.reshape(t.shape[0], -1, 3), e_uncond.reshape(t.shape[0], -1, 3)
        # two forwards, as the reference runs them (pair_cfg = False); forward() joins the side stream for the conditions
        x_t_sparse = self._adopt(x_t).sparse()
        e_cond = self.forward(x_t, x_t_sparse, x_cond, t)
        e_uncond = self.forward(x_t, x_t_sparse, x_uncond, t)
        return e_cond, e_uncond

    def denoise_step(self, x_init, x_t, x_cond, x_uncond, t_int: int, noise=None, parts=None, next_t=None):
        """One iteration of completion_loop (pipeline:158-167): CFG network pair, DPM-Solver++
        update on the per-point offsets, re-voxelisation of x_t and the two conditions (`parts`: the
        encoded conditions of a cache_condition run, which then skips their re-voxelisation too)."""
        t = torch.full((1,), t_int, dtype=torch.int64, device=self.device)
        e_cond, e_uncond = self.classfree_pair(x_t, x_cond, x_uncond, t, parts, t_host=t_int)
        x_t = self.step_boundary(x_init, x_t, e_cond, e_uncond, t_int, noise)
        if parts is None:
            x_cond, x_uncond = self.reset_partial_pcd(x_cond, x_uncond, next_t=next_t)
        return x_t, x_cond, x_uncond

    def step_boundary(self, x_init, x_t, e_cond, e_uncond, t_int: int, noise=None):
        """pipeline:153 + 161-164: guidance mix, offsets, dpm_scheduler.step, x_init + prev_sample, points_to_tensor -> the next
        step's field.  One launch (ops.cfg_dpm_step) when the inputs allow it, the torch sequence otherwise."""
        sch = self.dpm_scheduler
        if (self.fused_step and x_init.is_cuda and x_init.dtype == torch.float64 and x_init.dim() == 3
                and sch.algorithm_type == "sde-dpmsolver++" and x_t.F.dtype == torch.float32 and e_cond.dtype == torch.float32
                and (noise is None or (noise.dtype == torch.float64 and noise.is_cuda))):
            from . import ops
            if noise is None:                       # the draw dpm_scheduler.step makes (same generator state, same values)
                noise = torch.randn(x_init.shape, device=x_init.device, dtype=torch.float64)
            x0, feats, coords = ops.cfg_dpm_step(e_cond, e_uncond, self.w_uncond, x_t.F, x_init, sch.step_plan(t_int), noise,
                                                 self.hparams["data"]["resolution"], scale_batch_column=True)
            sch.commit(x0)
            return self._make_field(feats, coords, "x_t")
        noise_t = e_uncond + self.w_uncond * (e_cond - e_uncond)
        input_noise = x_t.F.reshape(x_init.shape[0], -1, 3) - x_init
        x_new = x_init + sch.step(noise_t, t_int, input_noise, noise=noise)["prev_sample"]
        return self.points_to_tensor(x_new, role="x_t")

    # pipeline:155-169
    def completion_loop(self, x_init, x_t, x_cond, x_uncond, noises=None):
        """_completion_loop_checked; in the opt-in two-piece fp16 mode (ops.SPLIT_PIECES = 2) a scan during which a value left fp16's
        range is redone from the same inputs, scheduler state and random draws on the default three bf16 pieces (a warning says
        so): the mode is never wrong, at worst slower."""
        from . import ops
        if not (ops.SPLIT_PIECES == 2 and x_t.F.device.type == "cuda"):
            return self._completion_loop_checked(x_init, x_t, x_cond, x_uncond, noises)
        sch = self.dpm_scheduler
        saved = {k: (list(v) if isinstance(v, list) else v) for k, v in sch.__dict__.items()}
        rng = torch.cuda.get_rng_state(self.device) if noises is None else None
        try:
            return self._completion_loop_checked(x_init, x_t, x_cond, x_uncond, noises)
        except ops.SplitRangeError as e:
            import warnings
            warnings.warn(f"two-piece fp16 scan voided ({e}); redone on three bf16 pieces")
        sch.__dict__.update({k: (list(v) if isinstance(v, list) else v) for k, v in saved.items()})
        if rng is not None:
            torch.cuda.set_rng_state(rng, self.device)
        with ops.split_pieces(3):
            return self._completion_loop_checked(x_init, x_t, x_cond, x_uncond, noises)

    def _completion_loop_checked(self, x_init, x_t, x_cond, x_uncond, noises=None):
        """The loop runs host-read-free where it can (read_free above); if the device later reports that an assumption of those
        steps did not hold (a tail map above its pair bound, a condition latent of another size than the step before), the whole
        loop is redone from the same inputs, scheduler state and random draws with exact sizes -- same results as if it had run
        that way from the start."""
        self.read_free_reset()          # the first pyramid of every role in this loop is built with its host read
        if not (self.read_free and x_t.F.device.type == "cuda"):
            out = self._completion_loop(x_init, x_t, x_cond, x_uncond, noises)
            if x_t.F.device.type == "cuda":
                from . import ops
                ops.split_check()                    # (two-piece fp16 mode only: a value beyond its range voids the scan, loudly)
            return out
        sch = self.dpm_scheduler
        saved = {k: (list(v) if isinstance(v, list) else v) for k, v in sch.__dict__.items()}
        rng = torch.cuda.get_rng_state(self.device) if noises is None else None
        try:
            out = self._completion_loop(x_init, x_t, x_cond, x_uncond, noises, check=False)
            why = self.read_free_check()
            from . import ops
            ops.split_check()                        # (two-piece fp16 mode only: a value beyond its range voids the scan, loudly)
        except RuntimeError as e:                    # (a consumer tripped over the same overflow before the host looked)
            if "bound" not in str(e):
                raise
            out, why = None, str(e)
        if why is None:
            x_t_last, out = out
            x_t_last.coordinate_manager.check()
            return out
        import warnings
        warnings.warn(f"host-read-free denoising loop voided ({why}); redone with exact sizes")
        sch.__dict__.update({k: (list(v) if isinstance(v, list) else v) for k, v in saved.items()})
        if rng is not None:
            torch.cuda.set_rng_state(rng, self.device)
        self.read_free_reset()
        prev, self.read_free = self.read_free, False
        try:
            return self._completion_loop(x_init, x_t, x_cond, x_uncond, noises)
        finally:
            self.read_free = prev

    def _completion_loop(self, x_init, x_t, x_cond, x_uncond, noises=None, check=True):
        parts = self.encode_conditions(x_cond, x_uncond) if self.cache_condition and self.pair_cfg else None
        ts = self.dpm_scheduler.host_timesteps
        for i, t_int in enumerate(ts):
            x_t, x_cond, x_uncond = self.denoise_step(x_init, x_t, x_cond, x_uncond, t_int,
                                                      None if noises is None else noises[i], parts,
                                                      next_t=ts[i + 1] if i + 1 < len(ts) else None)
        if not check:                    # (the caller validates the read-free steps before it trusts -- or reads -- anything)
            return x_t, x_t.F.cpu().detach().numpy()
        x_t.coordinate_manager.check()
        return x_t.F.cpu().detach().numpy()


def _merge(base, over):
    out = {k: (dict(v) if isinstance(v, dict) else v) for k, v in base.items()}
    for k, v in (over or {}).items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k].update(v)
        else:
            out[k] = v
    return out


def farthest_point_sample(points: torch.Tensor, n_samples: int) -> torch.Tensor:
    """open3d ``farthest_point_down_sample`` stand-in (pipeline:97-99): greedy FPS starting from index 0, returns
    the selected indices in selection order.  On the GPU this is the HIP kernel behind ``ops.farthest_point_sample``;
    CPU tensors (host-side tooling, golden-file generation) take the plain torch loop with the same arithmetic."""
    if points.is_cuda:
        from . import ops
        return ops.farthest_point_sample(points, n_samples)
    n = points.shape[0]
    if n_samples >= n:
        return torch.arange(n, device=points.device)
    sel = torch.empty(n_samples, dtype=torch.int64, device=points.device)
    dist = torch.full((n,), float("inf"), dtype=points.dtype, device=points.device)
    far = torch.zeros((), dtype=torch.int64, device=points.device)
    for i in range(n_samples):
        sel[i] = far
        d = ((points - points[far]) ** 2).sum(-1)
        dist = torch.minimum(dist, d)
        far = torch.argmax(dist)
    return sel


# ----------------------------------------------------------------------------------------
# point-cloud files (pipeline:171-177); own PLY reader instead of open3d
# ----------------------------------------------------------------------------------------
_PLY_TYPES = {"float": "f", "float32": "f", "double": "d", "float64": "d", "uchar": "B", "uint8": "B",
              "char": "b", "int8": "b", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
              "int": "i", "int32": "i", "uint": "I", "uint32": "I"}


def read_ply_points(path: str) -> np.ndarray:
    """x, y, z of the vertex element of an ASCII or binary-little-endian PLY file (float64)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list properties on vertices are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return np.stack([data[:, names.index(a)] for a in "xyz"], axis=1).astype(np.float64)
        if fmt != "binary_little_endian":
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        dt = np.dtype([(n, "<" + t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * n_vert), dtype=dt, count=n_vert)
        return np.stack([data[a].astype(np.float64) for a in "xyz"], axis=1)


def write_ply_points(path: str, points: np.ndarray):
    pts = np.asarray(points, dtype="<f8")
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {pts.shape[0]}\n"
                 "property double x\nproperty double y\nproperty double z\nend_header\n").encode())
        f.write(pts.tobytes())


def load_pcd(pcd_file):
    if pcd_file.endswith(".bin"):
        return np.fromfile(pcd_file, dtype=np.float32).reshape((-1, 4))[:, :3]
    if pcd_file.endswith(".ply"):
        return read_ply_points(pcd_file)
    raise ValueError(f"Point cloud format '.{pcd_file.split('.')[-1]}' not supported. "
                     "(supported formats: .bin (kitti format), .ply)")
