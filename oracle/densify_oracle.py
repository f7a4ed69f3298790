"""CPU restatement of the reference's adaptive density control -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference freegaussian/freegaussian_model.py line by line:
``refinement_after`` :404-491, ``cull_gaussians`` :493-522, ``split_gaussians`` :524-563,
``dup_gaussians`` :565-574, ``dup_in_optim`` :338-361, ``remove_from_optim`` :313-331, as pure
functions on dictionaries of CPU tensors (no nerfstudio, no optimizers objects: the Adam moments
are passed as ``{name: {"exp_avg": t, "exp_avg_sq": t}}``).  Only ``tests/`` may import this.

PARITY: PINNED to the reference's own methods.  The module cannot be imported here (its
nerfstudio / gsplat imports fail), so tests/golden/make_golden.py::gen_densify
executes the bodies of ``refinement_after`` / ``split_gaussians`` / ``dup_gaussians`` /
``cull_gaussians`` / ``dup_in_optim`` / ``remove_from_optim`` (AST slices, never stored) as methods
of a stub ``self`` on 8 seeded cases and tests/test_densify.py checks this file against their
outputs (tests/golden/g_densify.npz; all tensors exact, the rotated child offsets to 1e-6).  The
one external function they call, gsplat's ``quat_to_rotmat``, is not on disk: the stub is handed
the function below (its published formula, wxyz quaternion)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    w, x, y, z = torch.unbind(q, dim=-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(q.shape[:-1] + (3, 3))  # fmt: skip


def refinement_after(params: Dict[str, torch.Tensor], moments: Dict[str, Dict[str, torch.Tensor]], stats: dict,
                     cfg, step: int, num_train_data: int, samples: Optional[torch.Tensor] = None):
    """-> (new params, new moments, info).  ``stats``: xys_grad_norm, vis_counts, max_2Dsize, last_size.
    ``cfg``: any object with the reference's config field names."""
    params = {k: v.clone() for k, v in params.items()}
    moments = {k: {m: t.clone() for m, t in v.items()} for k, v in moments.items()}
    info = {"densified": 0, "culled_only": 0, "opacity_reset": 0, "n_splits": 0, "n_dups": 0}
    if step < cfg.refine_start:  # :406
        return params, moments, info
    max_2dsize = stats.get("max_2Dsize")
    reset_interval = cfg.reset_alpha_every * cfg.refine_every  # :413
    do_densification = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    deleted_mask = None
    if do_densification:
        avg_grad_norm = (stats["xys_grad_norm"] / stats["vis_counts"]) * 0.5 * max(stats["last_size"][0], stats["last_size"][1])
        high_grads = (avg_grad_norm > cfg.densify_grad_thresh).squeeze()
        splits = (params["scales"].exp().max(dim=-1).values > cfg.densify_size_thresh).squeeze()
        splits &= high_grads
        if step < cfg.stop_screen_size_at:
            splits |= (max_2dsize > cfg.split_screen_size).squeeze()
        nsamps = cfg.n_split_samples
        # split_gaussians :524-563
        n_splits = int(splits.sum().item())
        centered = torch.randn((nsamps * n_splits, 3)) if samples is None else samples
        scaled = torch.exp(params["scales"][splits].repeat(nsamps, 1)) * centered
        quats = params["quats"][splits] / params["quats"][splits].norm(dim=-1, keepdim=True)
        rots = quat_to_rotmat(quats.repeat(nsamps, 1))
        rotated = torch.bmm(rots, scaled[..., None]).squeeze(-1)
        split_params = {
            "means": rotated + params["means"][splits].repeat(nsamps, 1),
            "features_dc": params["features_dc"][splits].repeat(nsamps, 1),
            "features_rest": params["features_rest"][splits].repeat(nsamps, 1, 1),
            "opacities": params["opacities"][splits].repeat(nsamps, 1),
            "scales": torch.log(torch.exp(params["scales"][splits]) / 1.6).repeat(nsamps, 1),
            "quats": params["quats"][splits].repeat(nsamps, 1),
        }
        params["scales"][splits] = torch.log(torch.exp(params["scales"][splits]) / 1.6)  # :549, in place
        dups = (params["scales"].exp().max(dim=-1).values <= cfg.densify_size_thresh).squeeze()  # :430
        dups &= high_grads
        dup_params = {k: params[k][dups] for k in NAMES}  # :565-574
        for k in NAMES:
            params[k] = torch.cat([params[k], split_params[k], dup_params[k]], dim=0)
        n_dups = int(dups.sum().item())
        max_2dsize = torch.cat([max_2dsize, torch.zeros(nsamps * n_splits), torch.zeros(n_dups)], dim=0)
        for k in NAMES:  # dup_in_all_optim(split_idcs, nsamps) then (dup_idcs, 1)
            if k in moments and "exp_avg" in moments[k]:
                for m in ("exp_avg", "exp_avg_sq"):
                    t = moments[k][m]
                    t = torch.cat([t, torch.zeros_like(t[splits]).repeat((nsamps,) + (1,) * (t.dim() - 1))], dim=0)
                    t = torch.cat([t, torch.zeros_like(t[: dups.shape[0]][dups])], dim=0)
                    moments[k][m] = t
        splits_mask = torch.cat((splits, torch.zeros(nsamps * n_splits + n_dups, dtype=torch.bool)))
        deleted_mask = _cull(params, cfg, step, max_2dsize, splits_mask)
        info.update(densified=1, n_splits=n_splits, n_dups=n_dups)
    elif step >= cfg.stop_split_at and cfg.continue_cull_post_densification:
        deleted_mask = _cull(params, cfg, step, max_2dsize, None)
        info["culled_only"] = 1
    if deleted_mask is not None:  # remove_from_all_optim
        for k in NAMES:
            if k in moments and "exp_avg" in moments[k]:
                for m in ("exp_avg", "exp_avg_sq"):
                    moments[k][m] = moments[k][m][~deleted_mask]
    if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:  # :475-487
        reset_value = cfg.cull_alpha_thresh * 2.0
        params["opacities"] = torch.clamp(params["opacities"], max=torch.logit(torch.tensor(reset_value)).item())
        if "opacities" in moments and "exp_avg" in moments["opacities"]:
            for m in ("exp_avg", "exp_avg_sq"):
                moments["opacities"][m] = torch.zeros_like(moments["opacities"][m])
        info["opacity_reset"] = 1
    return params, moments, info


def _cull(params, cfg, step, max_2dsize, extra_cull_mask):
    """cull_gaussians :493-522 (mutates ``params``)."""
    culls = (torch.sigmoid(params["opacities"]) < cfg.cull_alpha_thresh).squeeze()
    if extra_cull_mask is not None:
        culls = culls | extra_cull_mask
    if step > cfg.refine_every * cfg.reset_alpha_every:
        toobigs = (torch.exp(params["scales"]).max(dim=-1).values > cfg.cull_scale_thresh).squeeze()
        if step < cfg.stop_screen_size_at and max_2dsize is not None:
            toobigs = toobigs | (max_2dsize > cfg.cull_screen_size).squeeze()
        culls = culls | toobigs
    for k in NAMES:
        params[k] = params[k][~culls]
    return culls
