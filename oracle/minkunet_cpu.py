"""CPU oracle for LiDiff's three sparse networks, evaluated from a ``state_dict``.

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see oracle/me_cpu.py).  Each function
follows /root/reference/lidiff/models/minkunet.py (line ranges cited) with the
MinkowskiEngine ops restated in oracle/me_cpu.py.  Eval-mode semantics only (BatchNorm
uses running statistics) -- this is the inference path of
diff_completion_pipeline.py:134-153.

The state-dict keys are the reference's (SURVEY.md 8b 'Checkpoint compatibility'), so a
``diff_net.ckpt`` / ``refine_net.ckpt`` state dict can be evaluated directly.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as TF

from . import me_cpu as me


_TRAIN = False


class train_mode:
    """``with train_mode():`` -- BatchNorm uses the statistics of the batch (nn.BatchNorm1d in training mode on the rows
    of F, eps 1e-5; Appendix A.8) instead of the running ones: the forward of DiffusionPoints.training_step
    (models.py:180-217).  The running statistics in `sd` are left untouched (the tests compare losses and gradients)."""

    def __enter__(self):
        global _TRAIN
        self.prev, _TRAIN = _TRAIN, True

    def __exit__(self, *exc):
        global _TRAIN
        _TRAIN = self.prev


def _bn(sd, p, x):
    if _TRAIN:
        return TF.batch_norm(x, None, None, sd[p + ".bn.weight"], sd[p + ".bn.bias"], training=True, eps=1e-5)
    return me.batch_norm_eval(x, sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                              sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"])


def _conv_bn_relu(sd, p, x, ks, stride, transpose=False):
    """BasicConvolutionBlock minkunet.py:13-29 / BasicDeconvolutionBlock :32-46."""
    fn = me.conv_transpose if transpose else me.conv
    y = fn(x, sd[p + ".net.0.kernel"], ks, stride)
    return y.replace(torch.relu(_bn(sd, p + ".net.1", y.F)))


def _residual(sd, p, x):
    """ResidualBlock minkunet.py:49-80."""
    y = me.conv(x, sd[p + ".net.0.kernel"], 3, 1)
    y = y.replace(torch.relu(_bn(sd, p + ".net.1", y.F)))
    y = me.conv(y, sd[p + ".net.3.kernel"], 3, 1)
    main = _bn(sd, p + ".net.4", y.F)
    if p + ".downsample.0.kernel" in sd:
        short = _bn(sd, p + ".downsample.1", me.conv_forward(x.F, sd[p + ".downsample.0.kernel"], None))
    else:
        short = x.F
    return y.replace(torch.relu(main + short))


def _stem(sd, p, x):
    """stem minkunet.py:93-100 / 155-162 / 511-518."""
    y = me.conv(x, sd[p + ".0.kernel"], 3, 1)
    y = y.replace(torch.relu(_bn(sd, p + ".1", y.F)))
    y = me.conv(y, sd[p + ".3.kernel"], 3, 1)
    return y.replace(torch.relu(_bn(sd, p + ".4", y.F)))


def _stage(sd, p, x):
    """stageN = conv2s2 block + 2 residual blocks (minkunet.py:102-124, 183-263, 520-542)."""
    y = _conv_bn_relu(sd, p + ".0", x, 2, 2)
    return _residual(sd, p + ".2", _residual(sd, p + ".1", y))


def _up(sd, p, x, skip):
    """upN = deconv2s2 block, ME.cat(skip), 2 residual blocks (minkunet.py:283-292,
    463-465; 544-572, 603-605)."""
    y = _conv_bn_relu(sd, p + ".0", x, 2, 2, transpose=True)
    assert y.ts == skip.ts
    y = y.replace(torch.cat([y.F, skip.F], dim=1))
    return _residual(sd, p + ".1.1", _residual(sd, p + ".1.0", y))


def _mlp(sd, p, x, act=lambda v: TF.leaky_relu(v, 0.1)):
    h = act(TF.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"]))
    return TF.linear(h, sd[p + ".2.weight"], sd[p + ".2.bias"])


def sub_dict(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def global_enc_forward(sd, field: me.CpuTensorField) -> me.CpuSparseTensor:
    """MinkGlobalEnc.forward minkunet.py:134-141."""
    x = _stem(sd, "stem", field.sparse())
    for n in (1, 2, 3, 4):
        x = _stage(sd, f"stage{n}", x)
    return x


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """MinkUNetDiff.get_timestep_embedding minkunet.py:390-401."""
    half = dim // 2
    freq = np.exp(np.arange(0, half) * -(np.log(10000) / (half - 1)))
    ang = t[:, None] * torch.from_numpy(freq).float()[None, :]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)
    return TF.pad(emb, (0, 1)) if dim % 2 == 1 else emb


def _condition(sd, name, x, part, temp_emb, order_tp=False):
    """One conditioning block of MinkUNetDiff.forward (e.g. minkunet.py:424-431):
    nearest part latent -> latent MLP; timestep MLP repeated per batch; fused MLP -> w;
    returns x * w."""
    idx = me.argmin_match(x.C, part.C)                      # :403-418
    p = _mlp(sd, f"latent_{name}", part.F[torch.from_numpy(idx)])
    t = _mlp(sd, f"{name}_temp", temp_emb)
    counts = np.unique(x.C[:, 0], return_counts=True)[1]    # :427 (rows are batch-grouped)
    t = torch.repeat_interleave(t, torch.from_numpy(counts), dim=0)
    w = _mlp(sd, f"latemp_{name}", torch.cat((t, p) if order_tp else (p, t), dim=-1))
    return x.replace(x.F * w)


def unet_diff_forward(sd, field: me.CpuTensorField, x_sparse: me.CpuSparseTensor,
                      part: me.CpuSparseTensor, t: torch.Tensor) -> torch.Tensor:
    """MinkUNetDiff.forward minkunet.py:420-497."""
    emb = timestep_embedding(t.float(), 96)
    x0 = _stem(sd, "stem", x_sparse)
    x1 = _stage(sd, "stage1", _condition(sd, "stage1", x0, part, emb))
    x2 = _stage(sd, "stage2", _condition(sd, "stage2", x1, part, emb))
    x3 = _stage(sd, "stage3", _condition(sd, "stage3", x2, part, emb))
    x4 = _stage(sd, "stage4", _condition(sd, "stage4", x3, part, emb))
    y1 = _up(sd, "up1", _condition(sd, "up1", x4, part, emb, order_tp=True), x3)   # :461
    y2 = _up(sd, "up2", _condition(sd, "up2", y1, part, emb), x2)
    y3 = _up(sd, "up3", _condition(sd, "up3", y2, part, emb), x1)
    y4 = _up(sd, "up4", _condition(sd, "up4", y3, part, emb), x0)
    sliced = y4.F[torch.from_numpy(field.inverse)]          # y4.slice(x).F  :497
    return _mlp(sd, "last", sliced)


def unet_refine_forward(sd, field: me.CpuTensorField) -> torch.Tensor:
    """MinkUNet.forward minkunet.py:596-619 (Tanh head :574-579)."""
    x0 = _stem(sd, "stem", field.sparse())
    x1 = _stage(sd, "stage1", x0)
    x2 = _stage(sd, "stage2", x1)
    x3 = _stage(sd, "stage3", x2)
    x4 = _stage(sd, "stage4", x3)
    y1 = _up(sd, "up1", x4, x3)
    y2 = _up(sd, "up2", y1, x2)
    y3 = _up(sd, "up3", y2, x1)
    y4 = _up(sd, "up4", y3, x0)
    return torch.tanh(_mlp(sd, "last", y4.F[torch.from_numpy(field.inverse)]))


def points_to_field(points: torch.Tensor, resolution=0.05, divide_batch_col=True) -> me.CpuTensorField:
    """DiffCompletion.points_to_tensor pipeline:68-84 (divides the batch column too,
    App. D.2) or DiffusionPoints.points_to_tensor models.py:162-178 (does not)."""
    feats = me.batched_coordinates(list(points), dtype=torch.float32)
    coord = feats.clone()
    if divide_batch_col:
        coord = torch.round(coord / resolution)
    else:
        coord[:, 1:] = torch.round(coord[:, 1:] / resolution)
    return me.CpuTensorField(feats[:, 1:].contiguous(), coord)


def denoise_forward(sd, x_field, x_sparse, cond_field, t):
    """DiffCompletion.forward pipeline:140-146: partial_enc(x_part) then model(...)."""
    part = global_enc_forward(sub_dict(sd, "partial_enc."), cond_field)
    out = unet_diff_forward(sub_dict(sd, "model."), x_field, x_sparse, part, t)
    return out.reshape(t.shape[0], -1, 3)


def classfree_forward(sd, x_field, cond_field, uncond_field, t, w=6.0):
    """DiffCompletion.classfree_forward pipeline:148-153."""
    xs = x_field.sparse()
    e_c = denoise_forward(sd, x_field, xs, cond_field, t)
    e_u = denoise_forward(sd, x_field, xs, uncond_field, t)
    return e_u + w * (e_c - e_u)



def training_loss(sd, pcd_full: torch.Tensor, pcd_part: torch.Tensor, noise: torch.Tensor, t: torch.Tensor,
                  drop_condition: bool = False, reg_weight: float = 5.0, beta_start=3.5e-5, beta_end=0.007, t_steps=1000):
    """DiffusionPoints.training_step (lidiff/models/models.py:180-217) as a function of the state dict, differentiable by
    torch autograd through the oracle's operators: t_sample = pcd_full + sqrt(1 - acp[t]) * noise (q_sample of zeros,
    :94-96,189), points_to_tensor without dividing the batch column (:162-178), single forward with the condition zeroed
    when `drop_condition` (:195-200), loss = mse + reg_weight * (mean^2 + (std - 1)^2) (:203-206).  Train-mode BatchNorm."""
    betas = torch.linspace(beta_start, beta_end, t_steps)                                    # scheduling.py:15-16
    acp = torch.tensor(np.cumprod((1.0 - betas).numpy(), axis=0), dtype=torch.float32)       # models.py:37-39
    t_sample = pcd_full + torch.sqrt(1.0 - acp)[t][:, None, None] * noise
    x_full = points_to_field(t_sample, divide_batch_col=False)
    x_part = points_to_field(torch.zeros_like(pcd_part) if drop_condition else pcd_part, divide_batch_col=False)
    with train_mode():
        denoise_t = denoise_forward(sd, x_full, x_full.sparse(), x_part, t)
    loss_mse = TF.mse_loss(denoise_t, noise)
    loss = loss_mse + reg_weight * (denoise_t.mean() ** 2 + (denoise_t.std() - 1.0) ** 2)
    return loss, denoise_t


def refine_training_loss(sd, pcd_noise: torch.Tensor, pcd_full: torch.Tensor, up_factor: int = 6):
    """RefineDiffusion.training_step (lidiff/models/models_refine.py:53-76) as a function of the state dict: voxelise the
    noisy cloud with the batch column divided too (:54-56, App. D.2), MinkUNet -> 6 offsets per point (:69-70), pytorch3d
    chamfer_distance defaults against pcd_full (:72: squared K=1 distances, mean over points, both directions added, batch
    mean).  The nearest-neighbour indices come from an exact KD-tree (oracle/metrics_cpu.py), the distances are re-formed
    from the matched rows in torch, so the loss is differentiable as pytorch3d's knn_gather formulation is."""
    from scipy.spatial import cKDTree
    field = points_to_field(pcd_noise, divide_batch_col=True)
    with train_mode():
        offset = unet_refine_forward(sd, field).reshape(-1, up_factor, 3)
    pred = (field.F[:, None, :] + offset).reshape(pcd_full.shape[0], -1, 3)
    total = pred.new_zeros(())
    for p, q in zip(pred, pcd_full.float()):
        j = torch.from_numpy(cKDTree(q.numpy().astype(np.float64)).query(p.detach().numpy().astype(np.float64))[1])
        i = torch.from_numpy(cKDTree(p.detach().numpy().astype(np.float64)).query(q.numpy().astype(np.float64))[1])
        total = total + (p - q[j]).square().sum(1).mean() + (q - p[i]).square().sum(1).mean()
    return total / pred.shape[0]
