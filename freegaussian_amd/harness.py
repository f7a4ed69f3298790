"""Minimal training harness around ``FreeGaussianModel`` (nerfstudio's Trainer is out of scope and
not installed): optimizer groups and schedules of the reference method spec, the reference loss
``0.8 L1 + 0.2 (1 - SSIM)`` (freegaussian_model.py:944-983), PSNR (:933), the per-step callback
order of SURVEY.md §3.1, and the view-DP exchange step.  Enough to run `ns-train freegaussian`'s
inner loop end to end on MI355X; dataset I/O and densification stay with the caller."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .method_config import STAGE1_OPTIMIZERS, OptimSpec
from .model import Camera, FreeGaussianModel


def _gauss_window(size: int = 11, sigma: float = 1.5, device=None) -> torch.Tensor:
    x = torch.arange(size, dtype=torch.float32, device=device) - size // 2
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    g = g / g.sum()
    return g[:, None] * g[None, :]


def ssim(a: torch.Tensor, b: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Mean SSIM of [B,C,H,W] images, 11x11 Gaussian window sigma 1.5, 'valid' borders -- the
    definition `pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)` uses."""
    C = a.shape[1]
    w = _gauss_window(device=a.device).to(a.dtype).expand(C, 1, 11, 11).contiguous()
    mu_a, mu_b = F.conv2d(a, w, groups=C), F.conv2d(b, w, groups=C)
    s_aa = F.conv2d(a * a, w, groups=C) - mu_a * mu_a
    s_bb = F.conv2d(b * b, w, groups=C) - mu_b * mu_b
    s_ab = F.conv2d(a * b, w, groups=C) - mu_a * mu_b
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (s_aa + s_bb + c2))
    return m.mean()


def psnr(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return -10.0 * torch.log10(F.mse_loss(pred, gt))


def l1_and_ssim(pred: torch.Tensor, gt: torch.Tensor):
    """(mean |gt - pred|, mean SSIM) of two [H,W,C] images: on the GPU the fused kernels (ops.l1_ssim: one launch each
    way instead of ~200, 10.4 ms -> 0.1 ms at 1080p); on the host the torch statement of the same definition (`ssim`),
    which is also what the GPU tests check the kernels against."""
    if pred.is_cuda:
        from . import ops

        return ops.l1_ssim(pred, gt)
    return (gt - pred).abs().mean(), ssim(gt.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None])


def main_loss(pred: torch.Tensor, gt: torch.Tensor, ssim_lambda: float = 0.2) -> torch.Tensor:
    """pred, gt: [H,W,3] in [0,1]."""
    l1, sim = l1_and_ssim(pred, gt)
    return (1 - ssim_lambda) * l1 + ssim_lambda * (1 - sim)


def build_optimizers(model: FreeGaussianModel, table: Optional[Dict[str, OptimSpec]] = None):
    table = table or STAGE1_OPTIMIZERS
    groups = model.get_param_groups()
    opts = {}
    for name, spec in table.items():
        if name in groups:
            # the Gaussian parameter groups (one big tensor each) on the GPU: the update in one launch per tensor
            # (optim.FusedAdam IS a torch.optim.Adam with step() replaced); the MLP groups -- many small tensors --
            # and anything on the host keep torch's multi-tensor step
            if name in model.gauss_params and all(p.is_cuda for p in groups[name]):
                from .optim import FusedAdam

                opts[name] = FusedAdam(groups[name], lr=spec.lr, eps=spec.eps)
            else:
                opts[name] = torch.optim.Adam(groups[name], lr=spec.lr, eps=spec.eps)
    return opts


def apply_schedules(opts, step: int, table: Optional[Dict[str, OptimSpec]] = None) -> None:
    """Exponential (log-linear) decay lr_init -> lr_final over max_steps, as nerfstudio's
    ExponentialDecayScheduler without warm-up."""
    table = table or STAGE1_OPTIMIZERS
    for name, opt in opts.items():
        spec = table[name]
        if spec.lr_final is None:
            continue
        t = min(max(step / spec.max_steps, 0.0), 1.0)
        lr = math.exp(math.log(spec.lr) * (1 - t) + math.log(spec.lr_final) * t)
        for g in opt.param_groups:
            g["lr"] = lr


def train_step(model: FreeGaussianModel, opts, camera: Camera, gt_image: torch.Tensor, step: int, table=None,
               grad_sync=None, num_train_data: Optional[int] = None, stats_sync=None, graphed=None,
               mask: Optional[torch.Tensor] = None, metrics_every: int = 1, dp=None) -> Dict[str, float]:  # fmt: skip
    """One iteration in the reference's callback order: step_cb -> get_outputs ->
    loss -> backward -> [view-DP gradient exchange] -> optimizers -> after_train_iter ->
    refinement_after every ``refine_every`` steps (freegaussian_model.py:575-590; needs
    ``num_train_data``, the number of training cameras, :416).  View-sharded DP: pass
    ``dp=viewdp.ModelViewDP(model)`` (the factored exchange: colour gradients all-gathered as 12-24 B per Gaussian from
    inside the backward, the rest one all-reduce of a flat buffer the gradients are views of) or
    ``grad_sync=viewdp.all_reduce_model_grads`` (one all-reduce of everything), and ``stats_sync=viewdp.sync_densify_stats``.
    ``graphed``: a ``graphed.GraphedModelStep(model)`` (its loss is the model's: ``main_loss`` with ``config.ssim_lambda``): while the scheduled resolution is
    launch-bound (the reference's first 6000 steps at 1/4 and 1/2 resolution) get_outputs + loss + backward
    replay as one hipGraph; the rest of the step is unchanged.  ``mask`` [H,W,1]: the batch's optional mask
    (freegaussian_model.py:956-963).  The loss is the sum of ``model.get_loss_dict`` (main loss + the optional
    scale regulariser), as nerfstudio's trainer sums it.  ``metrics_every`` = k: loss / psnr are read back (one host
    synchronisation) only on steps divisible by k, as a trainer that logs every k steps does; the other steps return
    the Gaussian count alone and the host runs ahead of the GPU."""
    model.step_cb(step)
    plain_loss = (mask is None and gt_image.shape[-1] == 3 and not model.config.use_scale_regularization
                  and not model.config.use_bilateral_grid)
    if graphed is not None and dp is None and plain_loss and graphed.applicable(camera):
        # (no zero_grad: the replay refills the .grad tensors, which are static buffers of the graph)
        out, loss = graphed.step(camera, gt_image)
        gt = graphed.static["gt"]
    else:
        if graphed is not None:
            graphed.release()  # grads of a graph that is no longer replayed must not be mistaken for fresh ones
        for o in opts.values():
            o.zero_grad(set_to_none=True)
        import contextlib

        with (dp.step() if dp is not None else contextlib.nullcontext()):  # (view-DP: the exchange runs on exit)
            out = model.get_outputs(camera)
            batch = {"image": gt_image} if mask is None else {"image": gt_image, "mask": mask}
            loss_dict = model.get_loss_dict(out, batch)
            loss = loss_dict["main_loss"] + loss_dict["scale_reg"]
            if "tv_loss" in loss_dict:  # (bilateral grids: freegaussian_model.py:988-989)
                loss = loss + loss_dict["tv_loss"]
            gt = None  # (composited below, only on the steps whose metrics are read)
            loss.backward()
    if grad_sync is not None:
        grad_sync(model)
    apply_schedules(opts, step, table)
    from .optim import step_all

    step_all(opts.values())  # (the FusedAdam groups in one launch, the others' own step())
    model.after_train_iter(step)
    if num_train_data is not None and step % model.config.refine_every == 0:
        from .densify import refinement_after

        if stats_sync is not None:  # view-DP: identical statistics and split samples on every rank
            stats_sync(model)
        refinement_after(model, opts, step, num_train_data)
    if metrics_every > 1 and step % metrics_every != 0:
        return {"gaussian_count": model.num_points}
    with torch.no_grad():
        if gt is None:
            gt = model.composite_with_background(model.get_gt_img(gt_image), out["background"])
        return {"loss": float(loss), "psnr": float(psnr(out["rgb"], gt)), "gaussian_count": model.num_points}
