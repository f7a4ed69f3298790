"""The reference's optional bilateral-grid colour correction (default off): ``use_bilateral_grid`` /
``color_corrected_metrics`` of ``freegaussian/freegaussian_model.py:122-126``, used at ``:227-233`` (one grid per
training image), ``:617-618`` (its parameter group), ``:879-882`` (applied to the rendered image while training),
``:935-937`` / ``:1024-1045`` (colour-corrected PSNR) and ``:988-989`` (10 x total variation of the grids).

The reference imports these four names from ``nerfstudio.model_components.lib_bilagrid`` -- a third-party dependency
(nerfstudio, pinned ``>= 1.1.3`` in the reference's pyproject) that is NOT in ``/root/reference`` and not installed
here.  This file restates that library's published algorithm ("Bilateral Guided Radiance Field Processing", Wang et
al. 2024: a per-image 3-D grid of 3x4 affine colour transforms, sliced at (x, y, luma) with trilinear interpolation)
with the same names, arguments and tensor layouts.  **Parity unpinned**: there is no golden vector for it in the
reference and the library cannot be run here; ``tests/test_bilagrid.py`` checks it against a scalar restatement of
trilinear slicing and against the properties the algorithm defines (identity grids, affine recovery).

Plain torch on purpose: an image-sized ``grid_sample`` plus a 3x4 multiply per pixel, off the section-8 hot path and
off by default -- it is here so a user of the reference who turns the switch on finds it."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn


def color_affine_transform(affine_mats: torch.Tensor, rgb: torch.Tensor) -> torch.Tensor:
    """``affine_mats`` [..., 3, 4] applied to ``rgb`` [..., 3]: A[:, :3] rgb + A[:, 3]."""
    return torch.matmul(affine_mats[..., :3], rgb.unsqueeze(-1)).squeeze(-1) + affine_mats[..., 3]


def total_variation_loss(x: torch.Tensor) -> torch.Tensor:
    """Mean squared forward difference of a batch of 3-D grids [B, C, L, H, W] along L, H and W, summed over the
    three axes and averaged over the batch."""
    tv = x.new_zeros(())
    for axis in (2, 3, 4):
        n = x.shape[axis]
        a = x.narrow(axis, 1, n - 1)
        b = x.narrow(axis, 0, n - 1)
        count = max(a[0].numel(), 1)
        tv = tv + (a - b).pow(2).sum() / count
    return tv / x.shape[0]


class BilateralGrid(nn.Module):
    """``num`` grids of 3x4 affine colour transforms, [num, 12, grid_W, grid_Y, grid_X], initialised to the identity.
    ``forward(grid_xy, rgb, idx)`` slices them at (x, y, luma(rgb)) and returns the per-pixel matrices [..., 3, 4]."""

    def __init__(self, num: int, grid_X: int = 16, grid_Y: int = 16, grid_W: int = 8):
        super().__init__()
        self.grid_width = grid_W
        eye = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0])
        self.grids = nn.Parameter(eye.view(1, 12, 1, 1, 1).repeat(num, 1, grid_W, grid_Y, grid_X))
        self.register_buffer("rgb2gray_weight", torch.tensor([[0.299, 0.587, 0.114]]))

    def rgb2gray(self, rgb: torch.Tensor) -> torch.Tensor:
        """Luma in [-1, 1] (the guidance coordinate of the slice)."""
        return (rgb @ self.rgb2gray_weight.T) * 2.0 - 1.0

    def tv_loss(self) -> torch.Tensor:
        return total_variation_loss(self.grids)

    def forward(self, grid_xy: torch.Tensor, rgb: torch.Tensor, idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``grid_xy`` [N, h, w, 2] in [0, 1]; ``rgb`` [N, h, w, 3]; ``idx`` [N] or [N, 1] picks the grids."""
        grids = self.grids
        if idx is not None:
            grids = grids[idx.reshape(-1)]
        assert grids.shape[0] == grid_xy.shape[0], (grids.shape, grid_xy.shape)
        lead = grid_xy.shape[:-1]
        xyz = torch.cat([(grid_xy - 0.5) * 2.0, self.rgb2gray(rgb)], dim=-1)  # [N,h,w,3] in [-1,1]: (x, y, luma)
        xyz = xyz.reshape(xyz.shape[0], 1, -1, 1, 3) if xyz.dim() != 4 else xyz.unsqueeze(1)  # [N,1,h,w,3]
        mats = F.grid_sample(grids, xyz, mode="bilinear", align_corners=True, padding_mode="border")  # [N,12,1,h,w]
        mats = mats.permute(0, 2, 3, 4, 1)
        return mats.reshape(*lead, 3, 4)


def slice(bil_grids: BilateralGrid, xy: torch.Tensor, rgb: torch.Tensor, grid_idx: torch.Tensor) -> Dict[str, torch.Tensor]:  # noqa: A001
    """(the library's name) Slice grid ``grid_idx`` at pixel coordinates ``xy`` in [0, 1] guided by ``rgb`` and apply
    the sliced transforms: {"rgb", "rgb_affine_mats"}.  Inputs without a batch axis get one."""
    shape = rgb.shape
    if xy.dim() == 2:  # [P,2] -> [1,P,1,2]
        xy, rgb = xy[None, :, None], rgb[None, :, None]
    elif xy.dim() == 3:  # [h,w,2]
        xy, rgb = xy[None], rgb[None]
    grid_idx = grid_idx.reshape(-1)
    if grid_idx.numel() == 1 and xy.shape[0] != 1:
        grid_idx = grid_idx.expand(xy.shape[0])
    mats = bil_grids(xy, rgb, grid_idx)
    out = color_affine_transform(mats, rgb)
    return {"rgb": out.reshape(shape), "rgb_affine_mats": mats.reshape(*shape[:-1], 3, 4)}


def apply_to_render(bil_grids: BilateralGrid, rgb: torch.Tensor, cam_idx: int, H: int, W: int) -> torch.Tensor:
    """What the reference's base class does with a rendered image [1, H, W, 3] of training camera ``cam_idx``
    (``_apply_bilateral_grid``, called at ``freegaussian_model.py:882``): a [0, 1]^2 pixel grid, one slice."""
    dev = rgb.device
    gy, gx = torch.meshgrid(torch.linspace(0, 1.0, H, device=dev), torch.linspace(0, 1.0, W, device=dev), indexing="ij")
    grid_xy = torch.stack([gx, gy], dim=-1).unsqueeze(0)
    return slice(bil_grids, grid_xy, rgb, torch.tensor([int(cam_idx)], device=dev, dtype=torch.long))["rgb"]


def color_correct(img: torch.Tensor, ref: torch.Tensor, num_iters: int = 5, eps: float = 0.5 / 255) -> torch.Tensor:
    """Warp ``img`` to match the colours of ``ref`` with a per-channel quadratic colour transform fitted by least
    squares on the pixels that are unclipped in both, re-fitted ``num_iters`` times (the metric-side correction of the
    bilateral-grid paper, after mip-NeRF 360's)."""
    if img.shape[-1] != ref.shape[-1]:
        raise ValueError(f"img's {img.shape[-1]} and ref's {ref.shape[-1]} channels must match")
    C = img.shape[-1]
    x = img.reshape(-1, C)
    y = ref.reshape(-1, C)

    def unclipped(z):
        return (z >= eps) & (z <= 1 - eps)

    mask0 = unclipped(x)
    for _ in range(num_iters):
        cols = [x[:, c : c + 1] * x[:, c:] for c in range(C)]  # the quadratic terms
        cols += [x, torch.ones_like(x[:, :1])]  # linear + bias
        A = torch.cat(cols, dim=-1)
        warp = []
        for c in range(C):
            b = y[:, c]
            m = mask0[:, c] & unclipped(x[:, c]) & unclipped(b)
            Am = torch.where(m[:, None], A, torch.zeros_like(A))
            bm = torch.where(m, b, torch.zeros_like(b))
            w = torch.linalg.lstsq(Am.double().cpu(), bm.double().cpu()[:, None]).solution[:, 0]
            assert bool(torch.isfinite(w).all())
            warp.append(w.to(x))
        x = torch.clamp(A @ torch.stack(warp, dim=-1), 0, 1)
    return x.reshape(img.shape)
