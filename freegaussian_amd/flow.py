"""Screen-space flow derivative: host side.

* ``relative_camera_motion`` -- (v, w) between two nerfstudio (OpenGL) camera poses with the
  conventions of the reference's ``diff_2d_epipolar_flow`` (preprocess/epipolar_flow.py:244-270):
  both poses are converted OpenGL->OpenCV incl. the world-axis swap, w = XYZ Euler angles of
  R0^-1 R1, v = t1 - t0.  Checked against the reference's own output (tests/golden/g_flow.npz).
* ``camera_flow_map`` -- the per-pixel camera flow  A v / Z + B w  (epipolar_flow.py:272-317) on
  the GPU (HIP kernel ``fg_camera_flow``).
* ``reprojection_motion`` / ``reprojection_flow_map`` -- F-spec', the exact-reprojection variant of
  the camera flow (preprocess/epipolar_flow_bp.py:258-298) with that file's own conventions;
  HIP kernel ``fg_reprojection_flow``.  To first order in the camera motion it equals
  A v / Z + B w with (v, w) = (translation, rotation vector) of the same 3x4 matrix
  (tests/test_host.py::test_reprojection_flow_agrees_with_the_AB_jacobian_to_first_order).
* ``flow_channels`` / ``render_with_flow`` -- F1 of SURVEY.md §8a: the composited Gaussian flow
  of Corollary 1 (docs/index.html:293-299), sum_i T_i alpha_i (mu_{i,t} - mu_{i,0}), rendered as
  two extra channels of the same raster pass, plus F2 (Lemma 1) per-Gaussian Jacobian terms
  through ``ops.gaussian_flow``.  The reference ships no loss that consumes these (SURVEY.md §0
  finding 3); the loss form here (L1 on finite-depth pixels) is a build choice."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops
from .rasterization import rasterization


def _opengl_to_opencv(c2w: torch.Tensor) -> torch.Tensor:
    """[3,4] or [4,4] camera-to-world, OpenGL -> OpenCV with the z-up/y-swap world change the
    reference applies (epipolar_flow.py:212-229, keep_original_world_coordinate=False)."""
    m = torch.eye(4, dtype=c2w.dtype, device=c2w.device)
    m[: c2w.shape[0]] = c2w
    m[2, :] = -m[2, :]
    m = m[[0, 2, 1, 3], :]
    m[0:3, 1:3] = -m[0:3, 1:3]
    return m


def _euler_xyz(R: torch.Tensor) -> torch.Tensor:
    """Extrinsic x-y-z Euler angles (scipy 'xyz'):  R = Rz(c) Ry(b) Rx(a)  ->  (a, b, c)."""
    b = -torch.asin(R[2, 0].clamp(-1.0, 1.0))
    a = torch.atan2(R[2, 1], R[2, 2])
    c = torch.atan2(R[1, 0], R[0, 0])
    return torch.stack([a, b, c])


def relative_camera_motion(c2w0: torch.Tensor, c2w1: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (veloc [3], omega [3]) in float64."""
    p0, p1 = _opengl_to_opencv(c2w0.double()), _opengl_to_opencv(c2w1.double())
    R_rel = torch.linalg.inv(p0[:3, :3]) @ p1[:3, :3]
    return p1[:3, 3] - p0[:3, 3], _euler_xyz(R_rel)


def camera_flow_map(depth: torch.Tensor, K: torch.Tensor, veloc: torch.Tensor, omega: torch.Tensor) -> torch.Tensor:
    """depth [H,W] or [H,W,1] (inf allowed) -> camera flow [H,W,2] (fp32, GPU)."""
    return ops.camera_flow(depth.reshape(depth.shape[0], depth.shape[1]), K, veloc, omega)


def reprojection_motion(c2w0: torch.Tensor, c2w1: torch.Tensor) -> torch.Tensor:
    """The 3x4 matrix the reference's reprojection applies to a camera-frame point of frame 0
    (epipolar_flow_bp.py:265-266, :275-276): both nerfstudio poses get the OpenGL->OpenCV
    camera-axis flip only (``manual2cv(keep_original_world_coordinate=True)``), then
    M = c2w1 . c2w0^-1.  float64."""

    def to_cv(c2w):
        m = torch.eye(4, dtype=torch.float64)
        m[: c2w.shape[0]] = c2w.double().cpu()
        m[0:3, 1:3] = -m[0:3, 1:3]
        return m

    return (to_cv(c2w1) @ torch.linalg.inv(to_cv(c2w0)))[:3]


def reprojection_flow_map(depth0: torch.Tensor, depth1: torch.Tensor, K: torch.Tensor, c2w0: torch.Tensor,
                          c2w1: torch.Tensor, opticalflow: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:  # fmt: skip
    """F-spec' on the GPU: ``sceneflow`` = -(uv - xy) and, with an optical-flow map,
    ``interflow`` = opticalflow - (uv - xy), both 0 at infinite depth -- the dictionary
    epipolar_flow_bp.diff_2d_epipolar_flow returns (:282-297).  depth maps [H,W] or [H,W,1]."""
    dev = depth0.device
    d0 = depth0.reshape(depth0.shape[0], depth0.shape[1])
    d1 = depth1.reshape(depth1.shape[0], depth1.shape[1])
    M = reprojection_motion(c2w0, c2w1).float().to(dev)
    scene = ops.reprojection_flow(d0, d1, K.to(dev), M, sign=-1.0)
    out = {"sceneflow": scene}
    if opticalflow is not None:
        inter = opticalflow.to(dev).float() + scene
        out["interflow"] = torch.where(torch.isinf(d0)[..., None], torch.zeros_like(inter), inter)
    return out


def render_with_flow(
    means_t, means_0, quats, scales, opacities, colors, viewmat_t, viewmat_0, K, width, height,
    sh_degree: Optional[int] = 3, render_mode: str = "RGB+ED", rasterize_mode: str = "classic",
) -> Dict[str, torch.Tensor]:  # fmt: skip
    """One raster pass that also composites the Gaussian flow (F1).

    ``means_t`` are the Gaussian centres at the current time under the current camera
    ``viewmat_t``; ``means_0`` the same Gaussians ``interval`` frames earlier under
    ``viewmat_0`` (the reference attaches that pose as ``cameras0``,
    freegaussian_model.py:769-770).  mu_0 comes from a second projection pass (K1 only)."""
    r0, mu0, _, _, _, _ = ops.project(means_0, quats, scales, viewmat_0[0], K[0], width, height)
    rt, mut, _, _, _, _ = ops.project(means_t, quats, scales, viewmat_t[0], K[0], width, height)
    both = ((r0 > 0) & (rt > 0)).unsqueeze(-1)
    disp = torch.where(both, mut - mu0, torch.zeros_like(mut))
    render, alpha, info = rasterization(
        means_t, quats, scales, opacities, colors, viewmat_t, K, width, height, sh_degree=sh_degree, packed=False,
        render_mode=render_mode, absgrad=True, rasterize_mode=rasterize_mode, extra_channels=disp,
    )  # fmt: skip
    n = render.shape[-1] - 2
    return {"render": render[..., :n], "flow_gs": render[..., n:], "alpha": alpha, "info": info}


def _pose_inverse_4x4(c2w: torch.Tensor) -> torch.Tensor:
    """to4x4(inverse(pose)) of nerfstudio.utils.poses for [...,3,4] poses."""
    R, t = c2w[..., :3, :3], c2w[..., :3, 3:]
    Rt = R.transpose(-1, -2)
    top = torch.cat([Rt, -Rt @ t], -1)
    bottom = torch.zeros_like(top[..., :1, :])
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], -2)


def query_3d_gaussian_flow(means2d, Z0, interflow, c2w1, K, grid_size: Optional[int] = None, step: int = 8,
                           reference_quirk: bool = True) -> Dict[str, torch.Tensor]:  # fmt: skip
    """F-dead of SURVEY.md section 8a: ``FreeGaussianModel.query_3d_gaussian_flow`` (``grid_size=None``,
    freegaussian_model.py:662-696) and ``query_3d_gaussian_flow_grid`` (``grid_size`` given,
    :698-751) -- never called upstream, kept for completeness of the flow surface.  Per on-screen
    Gaussian: sample ``interflow`` [1,H,W,2] at its ``means2d`` [1,N,2], advect, sample the depth
    map ``Z0`` [1,H,W,1] there, lift with K^-1 and map by ``inverse(c2w1)``; off-screen rows stay 0.
    -> {"p1_3d2": [1,N,3]} or [1,N,3*(4*(grid_size//2//step)^2+1)] for the grid variant.

    ``reference_quirk=True`` is the reference's arithmetic exactly: its floor/ceil ``bilinear_interp``
    (0 at integer coordinates, utils.py:316-343) and, in the grid variant, the height/width swap of
    ``_, w, h, _ = Z0.shape`` (:722: neighbour rows are clamped to W-1 and columns to H-1).  False:
    true bilinear sampling and the clamps on the right axes.  Plain torch, any device."""
    from .utils import bilinear_interp

    device = means2d.device
    _, H, W, _ = Z0.shape
    m = ((means2d >= 0) & (means2d < torch.tensor([W, H], device=device))).all(-1)  # x < W, y < H (:677)
    x, y = means2d[m][..., 0], means2d[m][..., 1]  # [n]
    if grid_size is None:
        x, y = x.unsqueeze(0), y.unsqueeze(0)  # [1,n]
    else:
        ylim, xlim = (W - 1, H - 1) if reference_quirk else (H - 1, W - 1)
        g = torch.arange(step, grid_size // 2 + 1, step, device=device)
        gy, gx = torch.meshgrid(torch.cat([-g, g]), torch.cat([-g, g]), indexing="ij")
        gy = (gy.reshape(-1, 1) + y).clamp(0, ylim).long()
        gx = (gx.reshape(-1, 1) + x).clamp(0, xlim).long()
        x, y = torch.cat([gx, x.unsqueeze(0)], 0), torch.cat([gy, y.unsqueeze(0)], 0)  # [B,n]
    B = x.shape[0]
    flow = bilinear_interp(interflow.float().expand(B, -1, -1, -1), x, y, reference_quirk=reference_quirk)
    x2, y2 = x + flow[..., 0], y + flow[..., 1]
    Z = bilinear_interp(Z0.expand(B, -1, -1, -1), x2, y2, reference_quirk=reference_quirk).permute(0, 2, 1)  # [B,1,n]
    p_cam1 = torch.linalg.inv(K) @ torch.stack([x2, y2, torch.ones_like(x2)], dim=1) * Z  # [B,3,n]
    p_homo = torch.cat([p_cam1, torch.ones_like(p_cam1[:, :1])], dim=1)
    p3d = (_pose_inverse_4x4(c2w1) @ p_homo)[:, :3]  # [B,3,n]
    N = means2d.shape[1]
    P1 = torch.zeros((1, N, 3 * B), device=device, dtype=p3d.dtype)
    P1[m] = p3d.permute(2, 0, 1).reshape(-1, 3 * B)
    return {"p1_3d2": P1}


def flow_loss(flow_gs: torch.Tensor, interflow: torch.Tensor, depth: Optional[torch.Tensor] = None) -> torch.Tensor:
    """L1 between the composited Gaussian flow and a target ``interflow`` map on pixels with
    finite depth (build choice, see module docstring)."""
    diff = (flow_gs - interflow).abs()
    if depth is not None:
        diff = diff[torch.isfinite(depth).expand_as(diff[..., :1]).squeeze(-1)]
    return diff.mean()
