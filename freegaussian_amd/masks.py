"""Per-Gaussian attribute masks for stage 2 (reference preprocess/knn_gaussian.py): render the
expected depth of a key frame, back-project the frame's 2-D attribute labels onto the Gaussians
whose centres are visible there (``fg_mask_backproject``), accumulate over key frames, save as
``gaussian_mask_NxM.npy`` -- the file ``freegaussian_pipeline.py:45-47`` loads.
SURVEY.md section 8f row 4."""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch

from . import _lib
from .rasterization import rasterization


def backproject_frame(gaussian_masks: torch.Tensor, means2d: torch.Tensor, depths: torch.Tensor, radii: torch.Tensor,
                      depth_map: torch.Tensor, atrb_masks: torch.Tensor, mask_valids: torch.Tensor) -> torch.Tensor:  # fmt: skip
    """In place: ``gaussian_masks [N,M]`` (bool, on the GPU) |= labels of one key frame
    (knn_gaussian.py:116-132).  ``means2d [N,2]``, ``depths [N]``, ``radii [N]`` are the unpacked
    ``info`` arrays of the frame's render (``radii > 0`` = the packed set), ``depth_map [H,W]`` its
    "ED" render, ``atrb_masks [H,W,M+1]`` / ``mask_valids [M+1]`` the frame's labels (the last one,
    background, is dropped: ``[..., :-1]``)."""
    if not gaussian_masks.is_cuda:
        raise _lib.FgRasterError("backproject_frame needs CUDA/HIP tensors: no CPU fallback")
    if gaussian_masks.dtype != torch.bool or not gaussian_masks.is_contiguous():
        raise ValueError("gaussian_masks must be a contiguous bool tensor [N,M]")
    N, M = gaussian_masks.shape
    H, W = depth_map.shape[-2:] if depth_map.dim() == 2 else depth_map.squeeze().shape
    dev = gaussian_masks.device
    labels = atrb_masks.to(device=dev, dtype=torch.bool).contiguous()
    valids = mask_valids.to(device=dev, dtype=torch.bool).contiguous()
    if labels.shape[:2] != (H, W) or labels.shape[2] != valids.shape[0] or labels.shape[2] != M + 1:
        raise ValueError(f"atrb_masks [H,W,M+1] / mask_valids [M+1] expected, got {tuple(labels.shape)}, {tuple(valids.shape)}")
    lib = _lib.load()
    _lib.check(lib.fg_mask_backproject(
        N, means2d.reshape(N, 2).float().contiguous().data_ptr(), depths.reshape(N).float().contiguous().data_ptr(),
        radii.reshape(N).to(torch.int32).contiguous().data_ptr(), depth_map.reshape(H, W).float().contiguous().data_ptr(),
        W, H, labels.data_ptr(), valids.data_ptr(), M + 1, M, gaussian_masks.data_ptr(),
        torch.cuda.current_stream().cuda_stream), "fg_mask_backproject")  # fmt: skip
    return gaussian_masks


@torch.no_grad()
def build_gaussian_masks(means, quats, scales, opacities, colors, sh_degree: Optional[int],
                         frames: Iterable[Tuple[torch.Tensor, torch.Tensor, int, int, torch.Tensor, torch.Tensor]],
                         rasterize_mode: str = "classic") -> torch.Tensor:  # fmt: skip
    """The key-frame loop of knn_gaussian.py:63-132 for activated parameters.  ``frames`` yields
    ``(viewmat [1,4,4], K [1,3,3], W, H, atrb_masks [H,W,M+1], mask_valids [M+1])``.
    -> bool [N,M] on the GPU."""
    out = None
    for viewmat, K, W, H, atrb_masks, mask_valids in frames:
        if out is None:
            out = torch.zeros(means.shape[0], mask_valids.shape[-1] - 1, dtype=torch.bool, device=means.device)
        render, _, info = rasterization(means, quats, scales, opacities, colors, viewmat.to(means.device),
                                        K.to(means.device), W, H, tile_size=16, packed=False, near_plane=0.01,
                                        far_plane=1e10, render_mode="ED", sh_degree=sh_degree, absgrad=False,
                                        rasterize_mode=rasterize_mode)  # fmt: skip
        backproject_frame(out, info["means2d"][0], info["depths"][0], info["radii"][0], render[0, ..., 0], atrb_masks,
                          mask_valids)  # fmt: skip
    if out is None:
        raise ValueError("no key frames")
    return out
