"""Adaptive density control: the reference's ``refinement_after`` with its split / duplicate /
cull steps and the Adam-state surgery (freegaussian/freegaussian_model.py:313-367, :404-571),
SURVEY.md section 8f row 2.

One implementation, HIP only (CPU tensors raise -- the torch restatement of the reference's op
sequence is test infrastructure: ``tests/densify_torch_sequence.py``, ``oracle/densify_oracle.py``):
``fg_densify_flags`` -> prefix sums -> ``fg_densify_map`` -> one ``fg_gather_rows`` per tensor
(6 parameters + 12 Adam moment tensors) straight into the final arrays -> ``fg_split_children``.
Every tensor is read once and written once; nothing is concatenated and re-masked.

The split samples are drawn with ``torch.randn((n_split_samples * n_splits, 3))`` exactly where
the reference does (:530), so replicas that share a seed (``viewdp.shared_seed``) stay identical."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from . import _lib

PARAM_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def _adam_state(opt: torch.optim.Optimizer):
    param = opt.param_groups[0]["params"][0]
    return param, opt.state.get(param, {})


def _swap_param(opt: torch.optim.Optimizer, new_param: nn.Parameter, new_state: dict) -> None:
    """What remove_from_optim / dup_in_optim do to the optimizer (:313-356)."""
    old = opt.param_groups[0]["params"][0]
    if old in opt.state:
        del opt.state[old]
    opt.param_groups[0]["params"] = [new_param]
    opt.state[new_param] = new_state


# ------------------------------------------------------------------------------------------------
# decisions shared by both paths


def _schedule(model, step: int, num_train_data: int):
    """-> (do_densify, do_cull_only, do_reset) as refinement_after decides them (:413-418, :466, :475)."""
    cfg = model.config
    reset_interval = cfg.reset_alpha_every * cfg.refine_every
    do_densify = step < cfg.stop_split_at and step % reset_interval > num_train_data + cfg.refine_every
    do_cull_only = (not do_densify) and step >= cfg.stop_split_at and cfg.continue_cull_post_densification
    do_reset = step < cfg.stop_split_at and step % reset_interval == cfg.refine_every
    return do_densify, do_cull_only, do_reset


def _reset_opacities(model, optimizers) -> None:
    """(:475-487)"""
    reset_value = model.config.cull_alpha_thresh * 2.0
    op = model.gauss_params["opacities"]
    op.data = torch.clamp(op.data, max=torch.logit(torch.tensor(reset_value)).item())
    opt = optimizers.get("opacities")
    if opt is not None:
        _, state = _adam_state(opt)
        if "exp_avg" in state:
            state["exp_avg"] = torch.zeros_like(state["exp_avg"])
            state["exp_avg_sq"] = torch.zeros_like(state["exp_avg_sq"])


# ------------------------------------------------------------------------------------------------
# HIP path


def _refine_fused(model, optimizers, step: int, do_densify: bool, samples: Optional[torch.Tensor]):
    lib = _lib.load()
    cfg = model.config
    gp = model.gauss_params
    dev = gp["means"].device
    N = gp["means"].shape[0]
    stream = torch.cuda.current_stream().cuda_stream

    def call(name, *args):
        _lib.check(getattr(lib, name)(*args), name)

    def ptr(t):
        return None if t is None else t.data_ptr()

    toobig_on = step > cfg.refine_every * cfg.reset_alpha_every
    screen_on = step < cfg.stop_screen_size_at
    m2d = model.max_2Dsize if model.max_2Dsize is not None else None
    flags = torch.empty(N, dtype=torch.uint8, device=dev)
    gn = model.xys_grad_norm.float().contiguous() if do_densify else None
    vc = model.vis_counts.float().contiguous() if do_densify else None
    call("fg_densify_flags", N, int(do_densify), float(max(model.last_size)), cfg.densify_grad_thresh,
         cfg.densify_size_thresh, cfg.split_screen_size if (do_densify and screen_on) else -1.0,
         cfg.cull_alpha_thresh, cfg.cull_scale_thresh if toobig_on else -1.0,
         cfg.cull_screen_size if (toobig_on and screen_on and m2d is not None) else -1.0,
         ptr(gn), ptr(vc), ptr(None if m2d is None else m2d.float().contiguous()),
         ptr(gp["scales"].detach().contiguous()), ptr(gp["opacities"].detach().contiguous()), ptr(flags), stream)  # fmt: skip
    # four exclusive prefix sums + totals: one cumsum over a [4,N] bit matrix, one 16-byte readback
    bits = torch.stack([(flags >> 2) & 1, (flags >> 3) & 1, (flags >> 4) & 1, flags & 1]).to(torch.int32)
    incl = torch.cumsum(bits, dim=1, dtype=torch.int32)
    excl = (incl - bits).contiguous()
    n_old, n_child, n_dup, n_split = (int(v) for v in incl[:, -1].tolist())
    nsamps = cfg.n_split_samples if do_densify else 0
    n_out = n_old + nsamps * n_child + n_dup
    src_index = torch.empty(n_out, dtype=torch.int32, device=dev)
    sample_index = torch.empty(n_out, dtype=torch.int32, device=dev)
    call("fg_densify_map", N, ptr(flags), ptr(excl[0]), ptr(excl[1]), ptr(excl[2]), ptr(excl[3]), n_old, n_child,
         n_split, nsamps, ptr(src_index), ptr(sample_index), stream)  # fmt: skip
    if do_densify:  # the reference's draw (:530), also when every child ends up culled
        z = torch.randn((nsamps * n_split, 3), device=dev) if samples is None else samples.to(dev).contiguous()
    new = {}
    for k in PARAM_NAMES:
        src = gp[k].detach().contiguous()
        D = src[0].numel() if N > 0 else 1
        dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
        call("fg_gather_rows", n_out, D, ptr(src), ptr(src_index), n_out, ptr(dst), stream)
        new[k] = dst
    # rows of split parents' children are moved and shrunk; with no split at all (duplicates only)
    # every new row is a plain copy and there is no sample buffer to hand over
    if n_out > n_old and n_split > 0:
        call("fg_split_children", n_old, n_out - n_old, ptr(sample_index), ptr(z), ptr(new["means"]),
             ptr(new["scales"]), ptr(new["quats"]), stream)  # fmt: skip
    for k in PARAM_NAMES:
        gp[k] = nn.Parameter(new[k])
        opt = optimizers.get(k)
        if opt is None:
            continue
        _, state = _adam_state(opt)
        if "exp_avg" in state:
            for m in ("exp_avg", "exp_avg_sq"):
                src = state[m].contiguous()
                dst = torch.empty((n_out,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
                call("fg_gather_rows", n_out, src[0].numel() if N > 0 else 1, ptr(src), ptr(src_index), n_old,
                     ptr(dst), stream)  # fmt: skip
                state[m] = dst
        _swap_param(opt, gp[k], state)
    return N, n_out


# ------------------------------------------------------------------------------------------------


def refinement_after(model, optimizers: Dict[str, torch.optim.Optimizer], step: int, num_train_data: int,
                     samples: Optional[torch.Tensor] = None, refine=None) -> Optional[Dict[str, int]]:  # fmt: skip
    """Mirror of FreeGaussianModel.refinement_after (:404-491).  ``optimizers`` maps the parameter
    group names (``means``, ``scales``, ...) to their single-parameter Adam optimizers, as
    nerfstudio's ``Optimizers.optimizers`` does.  ``samples`` overrides the randn draw (tests);
    ``refine`` replaces the HIP passes by another implementation of the split / duplicate / cull
    step with the same signature (tests and scripts/densify_bench.py time the reference's torch op
    sequence that way).  Returns counts, or None when nothing ran (before ``refine_start``)."""
    assert step == model.step
    cfg = model.config
    if step < cfg.refine_start:
        return None
    if refine is None:
        if not model.gauss_params["means"].is_cuda:
            raise _lib.FgRasterError("refinement_after needs CUDA/HIP tensors: the densification passes have no CPU fallback")
        refine = _refine_fused
    with torch.no_grad():
        do_densify, do_cull_only, do_reset = _schedule(model, step, num_train_data)
        before = after = model.num_points
        if do_densify:
            assert model.xys_grad_norm is not None and model.vis_counts is not None and model.max_2Dsize is not None
        if do_densify or do_cull_only:
            before, after = refine(model, optimizers, step, do_densify, samples)
        if do_reset:
            _reset_opacities(model, optimizers)
        model.xys_grad_norm = None
        model.vis_counts = None
        model.max_2Dsize = None
    return {"before": before, "after": after, "densified": int(do_densify), "culled_only": int(do_cull_only),
            "opacity_reset": int(do_reset)}  # fmt: skip
