"""The reference's densification op sequence in torch -- ``cat`` everything, then boolean-mask
everything (freegaussian/freegaussian_model.py:404-571: refinement_after, split_gaussians,
dup_gaussians, cull_gaussians and the optimizer surgery :313-367) -- written against the build's
model / optimizer objects.  TEST INFRASTRUCTURE: a second restatement next to
oracle/densify_oracle.py (which works on plain dicts); the product (freegaussian_amd/densify.py)
only has the HIP passes.  Pass ``refine=refine_torch`` to ``refinement_after`` to run it."""
from typing import Optional

import torch
from torch import nn

from freegaussian_amd.densify import PARAM_NAMES, _adam_state, _swap_param
from oracle.densify_oracle import quat_to_rotmat


def refine_torch(model, optimizers, step: int, do_densify: bool, samples: Optional[torch.Tensor]):
    cfg = model.config
    gp = model.gauss_params
    dev = gp["means"].device
    n0 = gp["means"].shape[0]
    extra_cull = None
    if do_densify:
        avg = (model.xys_grad_norm / model.vis_counts) * 0.5 * max(model.last_size[0], model.last_size[1])
        high = (avg > cfg.densify_grad_thresh).squeeze()
        splits = (gp["scales"].exp().max(dim=-1).values > cfg.densify_size_thresh).squeeze() & high
        if step < cfg.stop_screen_size_at:
            splits = splits | (model.max_2Dsize > cfg.split_screen_size).squeeze()
        nsamps = cfg.n_split_samples
        n_splits = int(splits.sum().item())
        # split_gaussians (:524-563)
        z = torch.randn((nsamps * n_splits, 3), device=dev) if samples is None else samples.to(dev)
        scaled = torch.exp(gp["scales"][splits].repeat(nsamps, 1)) * z
        q = gp["quats"][splits] / gp["quats"][splits].norm(dim=-1, keepdim=True)
        rots = quat_to_rotmat(q.repeat(nsamps, 1))
        new_means = torch.bmm(rots, scaled[..., None]).squeeze(-1) + gp["means"][splits].repeat(nsamps, 1)
        shrunk = torch.log(torch.exp(gp["scales"][splits]) / 1.6)
        split_params = {
            "means": new_means,
            "features_dc": gp["features_dc"][splits].repeat(nsamps, 1),
            "features_rest": gp["features_rest"][splits].repeat(nsamps, 1, 1),
            "opacities": gp["opacities"][splits].repeat(nsamps, 1),
            "scales": shrunk.repeat(nsamps, 1),
            "quats": gp["quats"][splits].repeat(nsamps, 1),
        }
        gp["scales"].data[splits] = shrunk  # in place, BEFORE `dups` is evaluated (:549, :430)
        dups = (gp["scales"].exp().max(dim=-1).values <= cfg.densify_size_thresh).squeeze() & high
        dup_params = {k: gp[k][dups] for k in PARAM_NAMES}
        for k in PARAM_NAMES:
            gp[k] = nn.Parameter(torch.cat([gp[k].detach(), split_params[k], dup_params[k]], dim=0))
        n_new = nsamps * n_splits + int(dups.sum().item())
        model.max_2Dsize = torch.cat([model.max_2Dsize, torch.zeros(n_new, device=dev)], dim=0)
        for k in PARAM_NAMES:  # dup_in_all_optim twice (:452-456)
            opt = optimizers.get(k)
            if opt is None:
                continue
            _, state = _adam_state(opt)
            if "exp_avg" in state:
                for m in ("exp_avg", "exp_avg_sq"):
                    state[m] = torch.cat([state[m], torch.zeros((n_new,) + state[m].shape[1:], device=dev)], dim=0)
            _swap_param(opt, gp[k], state)
        extra_cull = torch.cat([splits, torch.zeros(n_new, device=dev, dtype=torch.bool)])
    # cull_gaussians (:493-522)
    culls = (torch.sigmoid(gp["opacities"]) < cfg.cull_alpha_thresh).squeeze(-1)
    if extra_cull is not None:
        culls = culls | extra_cull
    if step > cfg.refine_every * cfg.reset_alpha_every:
        toobigs = (torch.exp(gp["scales"]).max(dim=-1).values > cfg.cull_scale_thresh).squeeze()
        if step < cfg.stop_screen_size_at and model.max_2Dsize is not None:
            toobigs = toobigs | (model.max_2Dsize > cfg.cull_screen_size).squeeze()
        culls = culls | toobigs
    for k in PARAM_NAMES:
        gp[k] = nn.Parameter(gp[k][~culls])
        opt = optimizers.get(k)
        if opt is None:
            continue
        _, state = _adam_state(opt)
        if "exp_avg" in state:
            state["exp_avg"] = state["exp_avg"][~culls]
            state["exp_avg_sq"] = state["exp_avg_sq"][~culls]
        _swap_param(opt, gp[k], state)
    return n0, int(gp["means"].shape[0])
