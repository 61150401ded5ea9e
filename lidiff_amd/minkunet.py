"""LiDiff's three sparse networks on the MI355X-native operator library.

Mirror of /root/reference/lidiff/models/minkunet.py -- same classes, constructor kwargs,
module tree and ``state_dict`` keys/shapes (so ``diff_net.ckpt`` / ``refine_net.ckpt`` load
unchanged), same forward semantics:

  BasicConvolutionBlock   minkunet.py:13-29      MinkGlobalEnc   minkunet.py:83-141
  BasicDeconvolutionBlock minkunet.py:32-46      MinkUNetDiff    minkunet.py:144-497
  ResidualBlock           minkunet.py:49-80      MinkUNet        minkunet.py:500-619

What is different is the execution plan in eval mode under ``torch.no_grad()`` (the default):
  * eval-mode BatchNorm, ReLU and the residual add are folded into the sparse-conv epilogue
    (one HBM round trip per conv instead of four);
  * ``ME.cat(y, skip)`` is never materialised: the following convs read two sources;
  * row-wise MLPs are applied BEFORE the row gather they commute with
    (``latent(part.F[idx]) == latent(part.F)[idx]``, ``last(y.slice(x).F) == last(y.F)[inv]``),
    and the first latemp Linear is split over its (p, t) inputs, so the per-voxel GEMM work is
    only the h x C_l projection;
  * the part->full nearest-voxel indices are computed once per coordinate map (decoder levels
    share the encoder's maps).
``with minkunet.fusion(False)`` (or training mode / grad enabled) runs the reference's op order
through the ME-API shim.
"""
from __future__ import annotations

import contextlib
import os


import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import MinkowskiEngine as ME
from . import ops

__all__ = ["MinkGlobalEnc", "MinkUNetDiff", "MinkUNet"]

CS = [32, 32, 64, 128, 256, 256, 128, 96, 96]


# ----------------------------------------------------------------------------------------
# fused conv + eval-BN (+ residual) (+ ReLU)
# ----------------------------------------------------------------------------------------
def _bn_affine(bn_mod: ME.MinkowskiBatchNorm):
    """scale/shift of an eval-mode BatchNorm1d, cached on the module until a tensor changes."""
    bn = bn_mod.bn
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.device)
    cache = getattr(bn_mod, "_affine_cache", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        cache = (key, scale, shift)
        bn_mod._affine_cache = cache
    return cache[1], cache[2]


_FUSION = True


class fusion:
    """``with minkunet.fusion(False):`` runs the reference's op order (every ME op its own launch)
    even in eval mode; the default fused plan is used only under ``torch.no_grad()`` in eval mode."""

    def __init__(self, enabled: bool):
        self.enabled = enabled

    def __enter__(self):
        global _FUSION
        self.prev, _FUSION = _FUSION, self.enabled

    def __exit__(self, *exc):
        global _FUSION
        _FUSION = self.prev


def _fusable(*mods) -> bool:
    return _FUSION and not torch.is_grad_enabled() and not any(m.training for m in mods)


# Sparse-conv tiles in Morton order of the output map (row_order of lidiff_spconv_fwd).  Measured on the bench
# workload it LOSES 5-10 % (spatially compact tiles have strongly varying pair counts -> worse load balance, and
# the L2 hits it buys do not pay for that), so it is off; the knob stays for maps with other statistics.
_ORDERED_TILES = False

# kernel_size-3 convolutions on low-density maps (the host's sparse-map hint) as centre pass + tail rows (ops.TailMap):
# the centre offset of a stride-1 map is the identity, the other offsets bring ~0.1-0.5 pairs per voxel
_CENTRE_TAIL = True

# transposed (kernel_size 2 / stride 2) convolutions with their output rows grouped by kernel offset (CoordinateManager.up_order):
# 128 pairs of ONE offset per tile instead of ~16 of each of the eight
_UP_ORDERED = True


_ME_BN = (ME.MinkowskiBatchNorm, ME.MinkowskiSyncBatchNorm)     # exact types whose .bn batch_norm_train() may stand in for


def _run_seq(seq, x: ME.SparseTensor) -> ME.SparseTensor:
    """seq(x) for a Sequential of ME modules -- module by module, as nn.Sequential does, except that in training a
    MinkowskiBatchNorm directly followed by a MinkowskiReLU runs as ONE normalise + ReLU pass (ops._BatchNormTrain(relu=True):
    no separate ReLU pass forward, its mask folded into the BatchNorm backward; same values)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if (i + 1 < len(mods) and type(m) in _ME_BN and isinstance(mods[i + 1], ME.MinkowskiReLU)
                and ops.bn_module_fused(m.bn) and ops.bn_fused_applies(m.bn, x.F)):
            x = x._like(ops.batch_norm_train(x.F, m.bn, relu=True))
            i += 2
            continue
        x = m(x)
        i += 1
    return x


def conv_bn_act(conv, bn, x: ME.SparseTensor, relu: bool, residual=None, extra=None) -> ME.SparseTensor:
    """relu?( bn(conv([x | extra])) + residual ) as ONE kernel launch (eval mode only)."""
    nbr, _, ts_out, _ = conv.maps(x, swapped=False)
    mgr = x.coordinate_manager
    m_out = mgr.maps[ts_out].coords.shape[0]
    # host-read-free maps (DiffCompletion.read_free): m_out is the BOUND of the rows, d_rows their count on the device; every
    # kernel CHOICE below is made from rows the host believes (mgr.rows: the same level of the role's previous pyramid)
    d_rows, free = mgr.count(ts_out), mgr.count(ts_out) is not None
    rows_out = mgr.rows(ts_out)
    hints = dict(d_rows=d_rows, rows_hint=rows_out, in_rows_hint=mgr.rows(x.tensor_stride)) if (free or mgr.hint_lag) else {}
    order = None
    if nbr is not None and _ORDERED_TILES:
        nbr, order = mgr.kernel_map_ordered(x.tensor_stride, ts_out, conv.kernel_size, conv.transposed)
    scale, shift = _bn_affine(bn)
    hint = conv.sparse_hint(x, ts_out)
    if conv.transposed and _UP_ORDERED and order is None:
        hit = mgr.up_order(x.tensor_stride, ts_out)
        if hit is not None:
            (nbr, order), hint = hit, False
            if ops.pairs_kernel_applies(x.F.shape[1], 0 if extra is None else extra.shape[1], conv.out_channels):
                pin, pout, off = mgr.up_pairs(x.tensor_stride, ts_out)       # one pair per output row: the streaming row kernel
                f = ops.spconv_fwd_pairs(x.F, conv.kernel, pin, pout, off, m_out, in_b=extra, scale=scale, shift=shift,
                                         residual=residual, relu=relu, replicas=x.replicas,
                                         rows_hint=hints.get("rows_hint"), in_rows_hint=hints.get("in_rows_hint"))
                out = ME.SparseTensor(f, tensor_stride=ts_out, coordinate_manager=mgr)
                out.replicas = x.replicas
                return out
    # centre + tail only on really isolated voxels (<= ~2 neighbours each: the 128-column rule of is_sparse_map); the wider
    # hint of the narrow tiles keeps the one-launch kernel with packed stages (3.6 neighbours per voxel: 429 vs 495 us)
    # (not for the 3-channel stem: the thin-input kernel walks the whole table in one launch)
    if (_CENTRE_TAIL and hint and conv.kernel_size == 3 and not conv.transposed and order is None and rows_out >= 1024
            and x.F.shape[1] > 4 and mgr.is_sparse_map(ts_out, ts_out, 3)
            # (shapes the pair-list kernel takes: every layer of the low-density levels; a wider one -- 192 -> 128 on a scene so
            #  small that even stride 4 is isolated voxels -- keeps the one-launch kernel, in every mode alike)
            and ops.pairs_kernel_applies(x.F.shape[1], 0 if extra is None else extra.shape[1], conv.out_channels)):
        if hints:
            hints["tail_hint"] = mgr.tail_rows(ts_out)
        f = ops.spconv_centre_tail(x.F, conv.kernel, mgr.tail_map(ts_out), m_out, in_b=extra, scale=scale, shift=shift,
                                   residual=residual, relu=relu, replicas=x.replicas, **hints)
    elif (conv.kernel_size == 3 and not conv.transposed and order is None and nbr is not None
          and ops.split3_layer(x.tensor_stride, rows_out, x.replicas, x.F.shape[1], 0 if extra is None else extra.shape[1],
                               conv.out_channels, m_bound=x.F.shape[0] // x.replicas)):
        # the dense levels: the contraction on the bf16 matrix pipe from three-way split operands (fp32 accuracy, ops.SPLIT3), the
        # map's rows sorted by their neighbour sets (whole 16-row blocks then lack an offset and are skipped; same bits).  The
        # output's own pieces are cut in the epilogue -- the next convolution of the level reads them
        if ops.SPLIT3_SORTED:
            nbr, order = mgr.kernel_map_mask_sorted(ts_out)
        f = ops.spconv_fwd_split3(x.F, conv.kernel, nbr, m_out, in_b=extra, scale=scale, shift=shift, residual=residual,
                                  relu=relu, replicas=x.replicas, d_rows=d_rows, want_planes=True, row_order=order)
    elif (conv.kernel_size == 1 and nbr is None and order is None and x.F.shape[1] + (0 if extra is None else extra.shape[1]) >= ops.SPLIT3_K1_MIN_CIN
          and ops.split3_layer(ts_out, rows_out, x.replicas, x.F.shape[1], 0 if extra is None else extra.shape[1], conv.out_channels,
                               m_bound=x.F.shape[0] // x.replicas)):
        # the widest 1 x 1 shortcut of the dense levels (384 -> 256 at stride 8: 423 -> 255 us) as a plain row GEMM on the same kernel; narrower
        # ones (192 -> 128: 154 vs 153 us) and the stride-2 up-convolutions (one neighbour per row, 8-stage tiles: 370 vs 316 us) measured no
        # better there and stay on the row / tile kernels
        f = ops.spconv_fwd_split3(x.F, conv.kernel, None, m_out, in_b=extra, scale=scale, shift=shift, residual=residual,
                                  relu=relu, replicas=x.replicas, d_rows=d_rows, want_planes=False)
    else:
        f = ops.spconv_fwd(x.F, conv.kernel, nbr, m_out, in_b=extra, scale=scale, shift=shift,
                           residual=residual, relu=relu, sparse_map=hint, replicas=x.replicas, row_order=order, **hints)
    out = ME.SparseTensor(f, tensor_stride=ts_out, coordinate_manager=mgr)
    out.replicas = x.replicas
    return out


# ----------------------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------------------
class BasicConvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, D=3):
        super().__init__()
        self.net = nn.Sequential(
            ME.MinkowskiConvolution(inc, outc, kernel_size=ks, dilation=dilation, stride=stride, dimension=D),
            ME.MinkowskiBatchNorm(outc), ME.MinkowskiReLU(inplace=True))

    def forward(self, x):
        if _fusable(self):
            return conv_bn_act(self.net[0], self.net[1], x, relu=True)
        return _run_seq(self.net, x)


class BasicDeconvolutionBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, D=3):
        super().__init__()
        self.net = nn.Sequential(
            ME.MinkowskiConvolutionTranspose(inc, outc, kernel_size=ks, stride=stride, dimension=D),
            ME.MinkowskiBatchNorm(outc), ME.MinkowskiReLU(inplace=True))

    def forward(self, x):
        if _fusable(self):
            return conv_bn_act(self.net[0], self.net[1], x, relu=True)
        return _run_seq(self.net, x)


class ResidualBlock(nn.Module):
    def __init__(self, inc, outc, ks=3, stride=1, dilation=1, D=3):
        super().__init__()
        conv = lambda a, b, k, s: ME.MinkowskiConvolution(a, b, kernel_size=k, dilation=dilation if k > 1 else 1,
                                                         stride=s, dimension=D)
        self.net = nn.Sequential(conv(inc, outc, ks, stride), ME.MinkowskiBatchNorm(outc),
                                 ME.MinkowskiReLU(inplace=True),
                                 conv(outc, outc, ks, 1), ME.MinkowskiBatchNorm(outc))
        self.downsample = nn.Sequential()
        if inc != outc or stride != 1:
            self.downsample = nn.Sequential(conv(inc, outc, 1, stride), ME.MinkowskiBatchNorm(outc))
        self.relu = ME.MinkowskiReLU(inplace=True)

    def forward(self, x, extra=None):
        """extra: second feature source on x's map, standing for ME.cat(x, extra)."""
        if _fusable(self):
            y = conv_bn_act(self.net[0], self.net[1], x, relu=True, extra=extra)
            if len(self.downsample):
                short = conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False, extra=extra).F
            else:
                short = x.F if extra is None else torch.cat([x.F, extra], dim=1)
            return conv_bn_act(self.net[3], self.net[4], y, relu=True, residual=short)
        if extra is not None:
            x = x._like(torch.cat([x.F, extra], dim=1))
        last = self.net[-1]
        if type(last) in _ME_BN and ops.bn_module_fused(last.bn):
            # training: the block's last BatchNorm, the shortcut add and the ReLU as one pass (ops._BatchNormTrain) -- decided
            # on the tensor that IS normalised (net[:-1]'s output) and only when the shortcut has its shape and dtype
            h = _run_seq(self.net[:-1], x)
            r = self.downsample(x)
            if ops.bn_fused_applies(last.bn, h.F) and r.F.shape == h.F.shape and r.F.dtype == h.F.dtype:
                return h._like(ops.batch_norm_train(h.F, last.bn, relu=True, residual=r.F))
            return self.relu(last(h) + r)
        return self.relu(_run_seq(self.net, x) + self.downsample(x))


def _stem(cin, c, D):
    return nn.Sequential(
        ME.MinkowskiConvolution(cin, c, kernel_size=3, stride=1, dimension=D), ME.MinkowskiBatchNorm(c),
        ME.MinkowskiReLU(True),
        ME.MinkowskiConvolution(c, c, kernel_size=3, stride=1, dimension=D), ME.MinkowskiBatchNorm(c),
        ME.MinkowskiReLU(inplace=True))


def _run_stem(stem, x):
    if _fusable(stem):
        return conv_bn_act(stem[3], stem[4], conv_bn_act(stem[0], stem[1], x, relu=True), relu=True)
    return _run_seq(stem, x)


def _stage(cin, cout, D):
    return nn.Sequential(BasicConvolutionBlock(cin, cin, ks=2, stride=2, dilation=1, D=D),
                         ResidualBlock(cin, cout, ks=3, stride=1, dilation=1, D=D),
                         ResidualBlock(cout, cout, ks=3, stride=1, dilation=1, D=D))


def _up(cin, cout, cskip, D):
    return nn.ModuleList([
        BasicDeconvolutionBlock(cin, cout, ks=2, stride=2, D=D),
        nn.Sequential(ResidualBlock(cout + cskip, cout, ks=3, stride=1, dilation=1, D=D),
                      ResidualBlock(cout, cout, ks=3, stride=1, dilation=1, D=D))])


def _run_up(up, x, skip):
    y = up[0](x)
    if _fusable(up):
        return up[1][1](up[1][0](y, extra=skip.F))
    return up[1](ME.cat(y, skip))


# training path: the conditioning MLPs' row-wise Linears in front of the gather (see MinkUNetDiff._condition)
_COMMUTE_TRAIN = True
# ... and the rest of the conditioning MLP on the (part row, batch) pair table (see MinkUNetDiff._condition)
_PAIR_TABLE_TRAIN = True


def _run_mlp(mlp, x):
    """A conditioning / head MLP on feature rows.  Under ops.train_operands("bf16") (BASELINE configs[4]) its Linears run as
    bf16 GEMMs with fp32 accumulation, forward and backward (torch.autocast), and hand fp32 rows back."""
    if ops.TRAIN_OPERANDS == "bf16" and torch.is_grad_enabled() and x.is_cuda:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return mlp(x).float()
    return mlp(x)


def _mlp(cin, hidden, cout, tail=None):
    layers = [nn.Linear(cin, hidden), nn.LeakyReLU(0.1, inplace=True), nn.Linear(hidden, cout)]
    if tail is not None:
        layers.append(tail)
    return nn.Sequential(*layers)


def _init_bn(module):
    for m in module.modules():
        if isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _Base(nn.Module):
    def _common(self, kwargs):
        cr = kwargs.get("cr", 1.0)
        self.cs = [int(cr * c) for c in CS]
        self.run_up = kwargs.get("run_up", True)
        self.D = kwargs.get("D", 3)
        return kwargs.get("in_channels", 3)

    def weight_initialization(self):
        _init_bn(self)

    def _encoder(self, cin):
        cs = self.cs
        self.stem = _stem(cin, cs[0], self.D)
        for n in range(4):
            setattr(self, f"stage{n + 1}", _stage(cs[n], cs[n + 1], self.D))


# ----------------------------------------------------------------------------------------
class MinkGlobalEnc(_Base):
    """Encoder of the partial scan -> stride-16, 256-channel latent (minkunet.py:83-141)."""

    def __init__(self, **kwargs):
        super().__init__()
        cin = self._common(kwargs)
        self.embed_dim = self.cs[-1]
        self._encoder(cin)
        self.weight_initialization()

    def forward(self, x):
        x = _run_stem(self.stem, x.sparse())
        for n in (1, 2, 3, 4):
            x = getattr(self, f"stage{n}")(x)
        return x.exact_view()      # (a no-op unless the field's maps were built without a host read: DiffCompletion.read_free)


# level name -> (channels multiplied by w, hidden width of latemp); order of minkunet.py:420-495
_LEVELS = ("stage1", "stage2", "stage3", "stage4", "up1", "up2", "up3", "up4")


class _RepeatSegments(torch.autograd.Function):
    """rows[b] repeated counts[b] times, batch after batch; backward: the column sum of every segment."""

    @staticmethod
    def forward(ctx, t, counts):
        ctx.counts = counts
        return torch.repeat_interleave(t, torch.tensor(counts, device=t.device), dim=0, output_size=sum(counts))

    @staticmethod
    def backward(ctx, g):
        out, lo = [], 0
        for c in ctx.counts:
            out.append(g[lo:lo + c].sum(dim=0))
            lo += c
        return torch.stack(out), None


class _BatchRows(torch.autograd.Function):
    """t[b] for every row of a coordinate map (b = the row's batch index): the same values as _RepeatSegments, from the
    batch column itself -- no host read of the rows per batch (torch.unique(...).tolist() is a device -> host sync per level
    and step).  Backward: the column sums per batch as ONE segment sum over the rows sorted by batch (ops.scatter_add_rows: fixed
    order, deterministic; the sort is cached on the batch-index tensor, which a level keeps for the step) -- rounds 2-4 ran a
    masked broadcast product and a reduction per batch, 4 ms of a 130 ms step."""

    @staticmethod
    def forward(ctx, t, bidx):
        ctx.save_for_backward(bidx)
        ctx.nb = t.shape[0]
        return t.index_select(0, bidx)

    @staticmethod
    def backward(ctx, g):
        (bidx,) = ctx.saved_tensors
        if g.is_cuda and g.dtype == torch.float32:
            return ops.scatter_add_rows(g, bidx, ctx.nb), None
        return torch.stack([(g * (bidx == b).unsqueeze(1)).sum(dim=0) for b in range(ctx.nb)]), None


class MinkUNetDiff(_Base):
    """The denoiser (minkunet.py:144-497)."""

    def __init__(self, **kwargs):
        super().__init__()
        cin = self._common(kwargs)
        cs = self.cs
        self.embed_dim = cs[-1]
        self.stem = _stem(cin, cs[0], self.D)
        lat, emb = cs[4], self.embed_dim
        # width of x at each conditioning point and hidden width of the fused MLP
        widths = {"stage1": (cs[0], cs[4]), "stage2": (cs[1], cs[4]), "stage3": (cs[2], cs[4]),
                  "stage4": (cs[3], cs[4]), "up1": (cs[4], cs[4]), "up2": (cs[5], cs[5]),
                  "up3": (cs[6], cs[6]), "up4": (cs[7], cs[7])}
        for i, name in enumerate(_LEVELS):
            cx, hid = widths[name]
            setattr(self, f"latent_{name}", _mlp(lat, lat, lat))
            setattr(self, f"latemp_{name}", _mlp(lat + lat, hid, cx))
            setattr(self, f"{name}_temp", _mlp(emb, emb, lat))
            if i < 4:
                setattr(self, name, _stage(cs[i], cs[i + 1], self.D))
            else:
                j = i - 4
                setattr(self, name, _up(cs[4 + j], cs[5 + j], cs[3 - j], self.D))
        self.last = _mlp(cs[8], 20, 3)
        self.weight_initialization()

    # -- minkunet.py:390-401 --------------------------------------------------------------
    def get_timestep_embedding(self, timesteps):
        assert timesteps.dim() == 1
        half = self.embed_dim // 2
        # the frequency table is a constant: kept on the device (a per-forward host -> device copy of it is a synchronous
        # pageable transfer -- the host then waits for the stream to drain before it can queue the network's launches:
        # ~0.5 ms of idle GPU per denoising step, profiles/r03_idle_gaps.txt)
        key = (str(timesteps.device), half)
        freq = self._freq_cache.get(key) if hasattr(self, "_freq_cache") else None
        if freq is None:
            f = np.exp(np.arange(0, half) * -(np.log(10000) / (half - 1)))
            freq = torch.from_numpy(f).float().to(timesteps.device)
            if not hasattr(self, "_freq_cache"):
                self._freq_cache = {}
            self._freq_cache[key] = freq
        emb = timesteps[:, None] * freq[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
        if self.embed_dim % 2 == 1:
            emb = TF.pad(emb, (0, 1), "constant", 0)
        return emb

    # -- minkunet.py:403-418 --------------------------------------------------------------
    def match_index(self, x_full, x_part, ahead: bool = False, by_batch: bool | None = None):
        """argmin_j ||C_full[i] - C_part[j]||^2 (batch column scaled by 2*max coord), cached per
        (full map, part tensor): decoder levels reuse the encoder's maps."""
        cache = x_full.coordinate_manager.aux
        key = ("match", x_full.tensor_stride, id(x_part.coordinate_manager), x_part.tensor_stride)
        hit = cache.get(key)
        if hit is not None and hit[0] is x_part.coordinate_manager:
            if len(hit) > 2 and hit[2] is not None:         # computed ahead on another stream (DiffusionPoints.training_step)
                cur = torch.cuda.current_stream(hit[1].device)
                cur.wait_event(hit[2])
                hit[1].record_stream(cur)
                cache[key] = hit = (hit[0], hit[1], None)
            return hit[1]
        # exhaustive scan: on the noisy x_t of the bench workload (sigma up to 1 m, many voxels far from every part voxel)
        # it beats the lattice-shell search of lidiff_nn_match_grid (0.44 vs 1.1 ms at 180k x 5.8k rows)
        x_full.coordinate_manager._acquire(x_full.tensor_stride)
        d_full = x_full.coordinate_manager.count(x_full.tensor_stride)
        # (training batches hold several scans: every row against its own batch element's part rows first -- the same indices)
        if by_batch is None:
            by_batch = torch.is_grad_enabled()
        idx = (ops.nn_match(x_full.C, x_part.C, by_batch=by_batch) if d_full is None
               else ops.nn_match_dev(x_full.C, d_full, x_part.C))
        done = None
        if ahead:                                              # the consumer's stream joins when it first asks (above)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(idx.device))
        cache[key] = (x_part.coordinate_manager, idx, done)    # the manager reference keeps the id unique
        return idx

    def match_part_to_full(self, x_full, x_part):
        return ME._GatherRows.apply(x_part.F, self.match_index(x_full, x_part))

    @staticmethod
    def _rows_per_batch(x):
        return torch.unique(x.C[:, 0], return_counts=True)[1]

    def _per_batch_rows(self, t, x):
        """minkunet.py:427-428 etc.: ``repeat_interleave(t, rows per batch)`` -- rows of a coordinate map are grouped
        by ascending batch index, so row r takes t[batch of r]: a gather through the map's batch column, without reading the
        rows per batch back to the host; the backward is one masked column sum per batch (torch's own backward of
        repeat_interleave is an index_add of M_l x C atomics into B rows: 5 ms per call at 360k rows)."""
        return _BatchRows.apply(t, self._batch_index(x))

    @staticmethod
    def _batch_index(x):
        aux = x.coordinate_manager.aux                  # lives and dies with the step's coordinate manager
        key = ("batch_index", x.tensor_stride)
        bidx = aux.get(key)
        if bidx is None:
            bidx = aux[key] = x.C[:, 0].long()
        return bidx

    def _condition_terms(self, name, part_feats, temp_emb):
        """The two summands of lin1(cat(latent(match), temp)) before the gather (fused plan): the row-wise MLPs run on
        the few part rows BEFORE the gather they commute with, and the first latemp Linear is split over its (p, t)
        inputs.  `part_feats` [M_p, 256] (the rows of several part tensors may be stacked).  Returns (h_p [M_p, h],
        h_t [B, h])."""
        latent, temp, latemp = (getattr(self, f"latent_{name}"), getattr(self, f"{name}_temp"),
                                getattr(self, f"latemp_{name}"))
        t_first = name == "up1"                      # minkunet.py:461: cat((t4, p4))
        lat = latent(part_feats)                                 # [M_p, 256] instead of [M_l, 256]
        lin1 = latemp[0]
        c = lat.shape[1]
        w_t, w_p = (lin1.weight[:, :c], lin1.weight[:, c:]) if t_first else (lin1.weight[:, c:], lin1.weight[:, :c])
        return lat @ w_p.t(), TF.linear(temp(temp_emb), w_t, lin1.bias)

    def _condition_table(self, name, parts, temp_emb):
        """w = latemp(cat(latent(part.F), temp)) evaluated on the PART rows of all replicas (one batch; see _condition)."""
        feats = parts[0].F if len(parts) == 1 else torch.cat([q.F for q in parts], dim=0)
        h_p, h_t = self._condition_terms(name, feats, temp_emb)
        return getattr(self, f"latemp_{name}")[2](TF.leaky_relu(h_p + h_t, 0.1))                  # [sum M_p, C]

    _tables = None

    def precompute_conditioning(self, part_feats, t):
        """The eight conditioning tables of a forward (fused plan, one batch) from the part latents and the timestep alone --
        ~50 small launches that need nothing of x.  DiffCompletion queues them right behind the condition encoders, BEFORE
        the host blocks on the map sizes of x_t: queued inside forward() they come when the host has no lead over the GPU
        and the device idles ~1 ms per step waiting for launches (profiles/r03_idle_gaps.txt).  Returns a dict for
        forward(..., cond=...); None when the plan does not apply."""
        parts = part_feats if isinstance(part_feats, (tuple, list)) else (part_feats,)
        if not _fusable(self) or t.shape[0] != 1:
            return None
        temp_emb = self.get_timestep_embedding(t)
        out = {"temp_emb": temp_emb, "parts": tuple(parts)}
        for name in _LEVELS:
            if getattr(self, f"latemp_{name}")[2].out_features % 4 == 0:
                out[name] = self._condition_table(name, parts, temp_emb)
        return out

    def _condition_hidden(self, name, x, part, temp_emb, out=None):
        """leaky(lin1(cat(latent(match), temp))) for the rows of x's coordinate map: gather + time bias + activation
        are one kernel."""
        h_p, h_t = self._condition_terms(name, part.F, temp_emb)
        idx = self.match_index(x, part)
        if h_t.shape[0] == 1 and h_p.shape[1] % 4 == 0:
            return ops.gather_bias_leaky(h_p, idx, h_t, 0.1, out=out)
        h_t = self._per_batch_rows(h_t, x)
        hidden = TF.leaky_relu(ops.gather_rows(h_p, idx) + h_t, 0.1)
        if out is not None:
            out.copy_(hidden)
            return out
        return hidden

    def _condition(self, name, x, part, temp_emb):
        """x * w with w = latemp(cat(latent(match), temp)) -- e.g. minkunet.py:424-431.  `part` may be a tuple of
        part tensors, one per replica of x (the CFG pair): their hidden layers fill one stacked buffer, so the
        second Linear runs once over all replicas (several batches), or -- one batch -- the whole MLP runs on the part
        rows and w is a gather of its output."""
        if _fusable(self):
            parts = part if isinstance(part, (tuple, list)) else (part,)
            assert len(parts) == x.replicas
            lin2 = getattr(self, f"latemp_{name}")[2]
            m = x.F.shape[0] // x.replicas
            out = torch.empty_like(x.F)
            if temp_emb.shape[0] == 1 and lin2.out_features % 4 == 0:
                # one batch: the time-embedding term is the same row everywhere, so the activation and the second
                # Linear are row-wise too and the WHOLE MLP commutes with the gather -- w = table[idx] with the
                # table evaluated on the part rows (one row for the single-voxel unconditional branch: broadcast)
                # (the part rows of all replicas go through the small MLPs together: one launch per Linear)
                tables = None if self._tables is None else self._tables.get(name)
                if tables is None:
                    tables = self._condition_table(name, parts, temp_emb)
                lo = 0
                for r, q in enumerate(parts):
                    table = tables[lo:lo + q.F.shape[0]]
                    lo += q.F.shape[0]
                    rows = slice(r * m, (r + 1) * m)
                    d_rows = x.coordinate_manager.count(x.tensor_stride)
                    if table.shape[0] == 1 and d_rows is None:
                        torch.mul(x.F[rows], table, out=out[rows])
                    else:           # (one part row: a broadcast of it -- the same products as torch.mul)
                        ops.gather_mul_rows(x.F[rows], table, None if table.shape[0] == 1 else self.match_index(x, q), out=out[rows],
                                            d_rows=d_rows)
                return x._like(out)
            hidden = torch.empty((x.F.shape[0], lin2.in_features), dtype=torch.float32, device=x.F.device)
            for r, q in enumerate(parts):
                self._condition_hidden(name, x, q, temp_emb, out=hidden[r * m:(r + 1) * m])
            torch.mul(x.F, lin2(hidden), out=out)
            return x._like(out)
        if _COMMUTE_TRAIN and torch.is_grad_enabled():
            # Training path (models.py:180-217), same arithmetic with the row-wise Linears moved in front of the gather they
            # commute with: latent(part.F[idx]) = latent(part.F)[idx], and lin1(cat(p, t)) = p W_p^T + (t W_t^T + b).  Of
            # the four Linears per level that the reference order runs over the M_l rows of x (360 000 at B = 2) only the last
            # one still does; the others see the few thousand part rows / the B time rows.  Autograd: the gather scatters
            # its gradient back onto the part rows, the per-batch broadcast sums its segments.
            lin2 = getattr(self, f"latemp_{name}")[2]
            amp = (torch.autocast("cuda", dtype=torch.bfloat16) if ops.TRAIN_OPERANDS == "bf16" and x.F.is_cuda
                   else contextlib.nullcontext())
            with amp:
                h_p, h_t = self._condition_terms(name, part.F, temp_emb)
            idx = self.match_index(x, part)
            nb, m_p = h_t.shape[0], h_p.shape[0]
            if _PAIR_TABLE_TRAIN and 2 * m_p * nb <= x.F.shape[0]:
                # Round 5: row i's weight depends on i only through the PAIR (matched part row idx[i], batch b(i)) -- at most
                # M_p x B distinct values (23 000 at B = 2) for the M_l rows of x (360 000 on the fine levels).  So the activation
                # and the second Linear run on the [M_p x B] pair table as well and w is a gather of it: the [M_l, 256] hidden
                # matrix of every level, its GEMM, the GEMM's weight gradient over K = M_l and a dozen elementwise passes over it
                # are gone.  Same function of the same inputs as the reference order (no assumption that a row's match lies in
                # its own batch element); autograd: the gather's segment sum, then the small MLP.
                hid = TF.leaky_relu(h_p.float().unsqueeze(1) + h_t.float().unsqueeze(0), 0.1)       # [M_p, B, h]
                table = _run_mlp(lin2, hid.reshape(m_p * nb, -1))                                     # [M_p B, C]
                aux = x.coordinate_manager.aux
                key = ("pair_index", x.tensor_stride, id(part.coordinate_manager), part.tensor_stride)
                comb = aux.get(key)
                if comb is None:                      # (kept for the step: the backward's sort by destination is cached on it)
                    comb = aux[key] = idx * nb + self._batch_index(x)
                if x.F.is_cuda and x.F.dtype == torch.float32 and table.dtype == torch.float32:
                    return x._like(ME._GatherMulRows.apply(x.F, table, comb))           # x * table[comb], one pass each way
                return x * ME._GatherRows.apply(table, comb)
            hidden = TF.leaky_relu(ME._GatherRows.apply(h_p.float(), idx) + self._per_batch_rows(h_t.float(), x), 0.1)
            return x * _run_mlp(lin2, hidden)
        latent, temp, latemp = (getattr(self, f"latent_{name}"), getattr(self, f"{name}_temp"),
                                getattr(self, f"latemp_{name}"))
        t_first = name == "up1"
        p = _run_mlp(latent, self.match_part_to_full(x, part))
        t = self._per_batch_rows(temp(temp_emb), x)
        return x * _run_mlp(latemp, torch.cat((t, p) if t_first else (p, t), -1))

    # -- minkunet.py:420-497 --------------------------------------------------------------
    def forward(self, x, x_sparse, part_feats, t, cond=None):
        """part_feats: the partial-scan latent, or (fused plan only) a tuple of R of them -- then the R conditioned
        forwards over the same x run as ONE stacked pass (same maps, same weights; every conv one launch with R
        replicas) and the result is a tuple of R per-point outputs.  That is the classifier-free-guidance pair of
        pipeline:148-153 / models.py:98-103 without running the network twice.
        cond: precompute_conditioning(part_feats, t) of exactly these latents and timestep (optional, same results)."""
        multi = isinstance(part_feats, (tuple, list))
        if multi and not _fusable(self):
            return tuple(self.forward(x, x_sparse, q, t) for q in part_feats)
        parts_now = tuple(part_feats) if multi else (part_feats,)
        use = cond is not None and _fusable(self) and len(cond["parts"]) == len(parts_now) and \
            all(a is b for a, b in zip(cond["parts"], parts_now))
        self._tables = cond if use else None
        try:
            return self._forward(x, x_sparse, part_feats, t, multi, cond["temp_emb"] if use else self.get_timestep_embedding(t))
        finally:
            self._tables = None

    MARK = None        # tools/step_timeline.py: callable(label) at a few points of the forward (records an event on the stream)

    def _forward(self, x, x_sparse, part_feats, t, multi, temp_emb):
        mark = self.MARK or (lambda label: None)
        mark("unet: enter")
        f0 = _run_stem(self.stem, x_sparse)                      # the stem sees no conditioning: shared
        mark("unet: stem")
        feats = [f0.replicate(len(part_feats)) if multi else f0]
        for name in _LEVELS[:4]:
            cond = self._condition(name, feats[-1], part_feats, temp_emb)
            mark(f"unet: {name} conditioned")
            feats.append(getattr(self, name)(cond))
            mark(f"unet: {name}")
        y = feats[4]
        for j, name in enumerate(_LEVELS[4:]):
            y = _run_up(getattr(self, name), self._condition(name, y, part_feats, temp_emb), feats[3 - j])
        if _fusable(self):
            inv = x.inverse_mapping
            if ops.slice_head_applies(self.last):
                # slice + head as one launch: a point reads its voxel's row and leaves three floats (head.hip) -- no [points, 96]
                # matrix, no N = 20 / N = 3 GEMMs; the sums do not depend on the voxel count or its bound
                out = ops.slice_head(y.F, inv, self.last, replicas=y.replicas if multi else 1)
            else:
                if multi:
                    m0 = y.F.shape[0] // y.replicas
                    inv = torch.cat([inv + r * m0 for r in range(y.replicas)])
                # (the head behind the slice, as the reference orders them: its GEMMs then see [points, 96] whatever the voxel
                #  count -- or its bound -- is, so the library picks the same kernels, i.e. the same summation order, in every mode)
                out = self.last(ops.gather_rows(y.F, inv))
            return tuple(out.chunk(y.replicas, dim=0)) if multi else out
        return _run_mlp(self.last, y.slice(x).F)


class MinkUNet(_Base):
    """The refinement network (minkunet.py:500-619): same UNet, no conditioning, Tanh head."""

    def __init__(self, **kwargs):
        super().__init__()
        cin = self._common(kwargs)
        cs = self.cs
        self._encoder(cin)
        for j in range(4):
            setattr(self, f"up{j + 1}", _up(cs[4 + j], cs[5 + j], cs[3 - j], self.D))
        self.last = _mlp(cs[8], 20, kwargs.get("out_channels", 3), tail=nn.Tanh())
        self.weight_initialization()
        self.dropout = nn.Dropout(0.3, True)

    def forward(self, x):
        feats = [_run_stem(self.stem, x.sparse())]
        for n in (1, 2, 3, 4):
            feats.append(getattr(self, f"stage{n}")(feats[-1]))
        y = feats[4]
        for j in range(4):
            y = _run_up(getattr(self, f"up{j + 1}"), y, feats[3 - j])
        if _fusable(self):
            return ops.gather_rows(self.last(y.F), x.inverse_mapping)
        return _run_mlp(self.last, y.slice(x).F)
