"""Host-side mirror of the reference's model surface around the raster call.

``FreeGaussianModel.get_outputs(camera)`` here reproduces rows H1-H4 + O1 of SURVEY.md §8a --
the ~50 lines of reference ``freegaussian/freegaussian_model.py:753-898`` that assemble the
inputs of ``rasterization(...)`` and post-process its outputs -- and S1 (``after_train_iter``,
:369-392).  ``FreeGaussianControlModel.get_outputs`` mirrors the stage-2 variant
(``freegaussian_control_model.py:52-209``).  nerfstudio is not a dependency: ``Camera`` is a
minimal stand-in for the fields of ``nerfstudio.cameras.Cameras`` the reference touches, and
``nerfstudio_adapter.py`` registers the methods only if nerfstudio imports.

Everything here is plain torch (small elementwise ops and the MLP GEMMs); the raster itself is
the HIP library behind ``freegaussian_amd.rasterization``."""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Union

import torch
from torch import nn

from .deform import FreeGaussianControllableModel, FreeGaussianDeformableModel
from .rasterization import num_sh_bases, rasterization
from .utils import get_viewmat, knn_mean_distance, random_quat_tensor, resize_image, transform_points


@dataclass
class Camera:
    """One pinhole camera (the reference asserts a batch of exactly one, :773)."""

    camera_to_worlds: torch.Tensor  # [1,3,4] OpenGL convention (nerfstudio)
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int
    times: Optional[torch.Tensor] = None  # [1,1]
    metadata: Dict = field(default_factory=dict)

    @property
    def shape(self):
        return (self.camera_to_worlds.shape[0],)

    def rescale_output_resolution(self, s: float) -> None:
        """In place, like nerfstudio's Cameras.rescale_output_resolution with its default
        ``scale_rounding_mode="floor"``: intrinsics scale linearly, the image size is floored --
        the same ``H // d`` the ground truth gets from ``resize_image`` (utils.py:248-261), so
        render and target agree in shape at every size (1014 rows at d = 4 -> 253, not 254).
        (nerfstudio's source is not readable here: SURVEY.md citation discipline -- the default is
        recalled; the shape agreement with ``get_gt_img`` is what the tests pin.)"""
        self.fx, self.fy, self.cx, self.cy = self.fx * s, self.fy * s, self.cx * s, self.cy * s
        # s is 1 / 2^k or 2^k: the product is exact in floating point, floor is safe
        self.width = int(self.width * s)
        self.height = int(self.height * s)

    def get_intrinsics_matrices(self) -> torch.Tensor:
        return torch.tensor([[[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]]])


@dataclass
class OrientedBox:
    """Stand-in for ``nerfstudio.data.scene_box.OrientedBox`` as ``set_crop`` / ``get_outputs`` use
    it (:397-398, :779-783): any object with ``within(points[N,3]) -> bool[N] or [N,1]`` works.
    R [3,3] rotation, T [3] centre, S [3] full side lengths (recalled layout; nerfstudio's source is
    not readable here)."""

    R: torch.Tensor
    T: torch.Tensor
    S: torch.Tensor

    def within(self, pts: torch.Tensor) -> torch.Tensor:
        R, T, S = (x.to(pts) for x in (self.R, self.T, self.S))
        local = (pts - T) @ R  # = R^T (p - T) per row
        return ((local > -S / 2) & (local < S / 2)).all(dim=-1, keepdim=True)


@dataclass
class FreeGaussianModelConfig:
    """Raster-relevant subset of reference FreeGaussianModelConfig (:51-131), same names/defaults."""

    warm_up: int = 3000
    refine_every: int = 100
    resolution_schedule: int = 3000
    background_color: str = "random"
    num_downscales: int = 2
    sh_degree_interval: int = 1000
    sh_degree: int = 3
    stop_split_at: int = 15000
    # adaptive density control (:58-88), consumed by freegaussian_amd.densify.refinement_after
    refine_start: int = 500
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0008
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    output_depth_during_training: bool = False
    ssim_lambda: float = 0.2  # (:96)
    use_scale_regularization: bool = False  # (:102; the PhysGaussian ratio penalty, every 10th step)
    max_gauss_ratio: float = 10.0  # (:104)
    use_bilateral_grid: bool = False  # (:122; per-image affine colour grids, freegaussian_amd/bilagrid.py)
    grid_shape: tuple = (16, 16, 8)  # (:124; X, Y, W)
    color_corrected_metrics: bool = False  # (:126)
    rasterize_mode: str = "classic"
    num_random: int = 50000
    random_scale: float = 10.0
    # build extension: fold the SH cat, exp / sigmoid / quaternion
    # normalisation (+ MLP deltas) and the background composite + clamp into the HIP passes
    fused_front_end: bool = True


_POSE_RING = None


def _pose_slot():
    """(tensor[25], its numpy view) of the next slot of a ring of pinned staging buffers for viewmat + K."""
    global _POSE_RING
    if _POSE_RING is None:
        ring = torch.empty(64, 32, dtype=torch.float32).pin_memory()
        _POSE_RING = [ring, ring.numpy(), 0]
    ring, ring_np, i = _POSE_RING
    _POSE_RING[2] = (i + 1) % ring.shape[0]
    return ring[i, :25], ring_np[i, :25]


# Per-step bookkeeping attributes (plain tensors / ints, never Parameters or sub-modules) are set past
# nn.Module.__setattr__, whose parameter / buffer / module checks cost ~5 us a time, six times a step.
_plain_set = object.__setattr__


class FreeGaussianModel(nn.Module):
    """Gaussian parameter store + deform/control MLPs + ``get_outputs``."""

    def __init__(self, config: Optional[FreeGaussianModelConfig] = None, num_points: Optional[int] = None,
                 seed_points: Optional[torch.Tensor] = None, is_blender: bool = True,
                 init_scales: Optional[float] = None, num_train_data: Optional[int] = None):  # fmt: skip
        """``num_train_data``: the number of training images (one bilateral grid each, :227-233).  ``init_scales``: None = the reference's initial scales, the mean distance to the three nearest neighbours
        (:158-162; a tree query over all points: a minute at 1M); a float = that log-scale everywhere, for harnesses that
        overwrite the scales anyway."""
        super().__init__()
        self.config = config or FreeGaussianModelConfig()
        if seed_points is not None:
            means = seed_points.float()
        else:
            n = num_points or self.config.num_random
            means = (torch.rand(n, 3) - 0.5) * self.config.random_scale
        n = means.shape[0]
        dim_sh = num_sh_bases(self.config.sh_degree)
        # layout and activations of reference gauss_params (:187-196): log-scales, logit-opacities
        self.gauss_params = nn.ParameterDict(
            {
                "means": nn.Parameter(means),
                "scales": nn.Parameter(torch.log(knn_mean_distance(means, 3)).repeat(1, 3) if init_scales is None
                                       else torch.full((n, 3), float(init_scales))),
                "quats": nn.Parameter(random_quat_tensor(n)),
                "features_dc": nn.Parameter(torch.rand(n, 3)),
                "features_rest": nn.Parameter(torch.zeros(n, dim_sh - 1, 3)),
                "opacities": nn.Parameter(torch.logit(0.1 * torch.ones(n, 1))),
            }
        )
        self.deform = FreeGaussianDeformableModel(is_blender=is_blender)  # :198
        self.control = FreeGaussianControllableModel()  # :200
        if self.config.use_bilateral_grid:  # (:227-233)
            from .bilagrid import BilateralGrid

            if not num_train_data:
                raise ValueError("use_bilateral_grid needs num_train_data (one grid per training image)")
            gx, gy, gw = self.config.grid_shape
            self.bil_grids = BilateralGrid(num=int(num_train_data), grid_X=gx, grid_Y=gy, grid_W=gw)
        self.step = 0
        self.background_color = torch.zeros(3)
        self.xys: Optional[torch.Tensor] = None
        self.radii: Optional[torch.Tensor] = None
        self.xys_grad_norm: Optional[torch.Tensor] = None
        self.vis_counts: Optional[torch.Tensor] = None
        self.max_2Dsize: Optional[torch.Tensor] = None
        self.last_size = (1, 1)
        self.last_list_length = 0
        self.crop_box = None  # (:220) set through set_crop by the viewer / render scripts
        self._active_crop: Optional[torch.Tensor] = None

    def set_crop(self, crop_box) -> None:
        """(:397-398)"""
        self.crop_box = crop_box

    @staticmethod
    def get_empty_outputs(width: int, height: int, background: torch.Tensor):
        """(:641-646) what an eval render returns when the crop box holds no Gaussian."""
        rgb = background.repeat(height, width, 1)
        depth = background.new_ones(*rgb.shape[:2], 1) * 10
        accumulation = background.new_zeros(*rgb.shape[:2], 1)
        return {"rgb": rgb, "depth": depth, "accumulation": accumulation, "background": background}

    def _crop_ids(self):
        """Eval-only crop (:778-785): None = all Gaussians; an all-False mask = empty output."""
        if self.crop_box is None or self.training:
            return None
        return self.crop_box.within(self.gauss_params["means"]).reshape(-1)

    # -- accessors with the reference's names -------------------------------------------------
    def _p(self, name: str) -> torch.Tensor:
        """A Gaussian parameter; inside a cropped eval render, its rows within the crop box (the
        reference's ``*_crop`` tensors, :786-798)."""
        t = self.gauss_params[name]
        return t if self._active_crop is None else t[self._active_crop]

    means = property(lambda self: self._p("means"))
    scales = property(lambda self: self._p("scales"))
    quats = property(lambda self: self._p("quats"))
    features_dc = property(lambda self: self._p("features_dc"))
    features_rest = property(lambda self: self._p("features_rest"))
    opacities = property(lambda self: self._p("opacities"))
    num_points = property(lambda self: self.gauss_params["means"].shape[0])
    device = property(lambda self: self.gauss_params["means"].device)

    def load_state_dict(self, state_dict, **kwargs):  # type: ignore[override]
        """(:278-291) the Gaussian parameters are re-allocated to the checkpoint's count, legacy
        un-prefixed names are remapped, and -- as in the reference -- the step jumps to 30000 (SH
        degree maxed, warm-up over)."""
        self.step = 30000
        state_dict = dict(state_dict)
        if "means" in state_dict:
            for p in ("means", "scales", "quats", "features_dc", "features_rest", "opacities"):
                state_dict[f"gauss_params.{p}"] = state_dict.pop(p)
        if "gauss_params.means" in state_dict:
            newp = state_dict["gauss_params.means"].shape[0]
            for name, param in self.gauss_params.items():
                new_shape = (newp,) + tuple(param.shape[1:])
                if tuple(param.shape) != new_shape:
                    self.gauss_params[name] = nn.Parameter(torch.zeros(new_shape, device=param.device))
        return super().load_state_dict(state_dict, **kwargs)

    # -- H3 ---------------------------------------------------------------------------------------
    def _get_downscale_factor(self) -> int:
        """2^max(num_downscales - step // resolution_schedule, 0) while training, else 1 (:626-633)."""
        if self.training:
            return 2 ** max(self.config.num_downscales - self.step // self.config.resolution_schedule, 0)
        return 1

    def _get_background_color(self) -> torch.Tensor:
        """(:648-660)"""
        mode = self.config.background_color
        if mode == "random":
            return torch.rand(3, device=self.device) if self.training else self.background_color.to(self.device)
        if mode == "white":
            return torch.ones(3, device=self.device)
        if mode == "black":
            return torch.zeros(3, device=self.device)
        raise ValueError(f"Unknown background color {mode}")

    def get_gt_img(self, image: torch.Tensor) -> torch.Tensor:
        """(:900-909)"""
        if image.dtype == torch.uint8:
            image = image.float() / 255.0
        d = self._get_downscale_factor()
        return (resize_image(image, d) if d > 1 else image).to(self.device)

    def composite_with_background(self, image: torch.Tensor, background: torch.Tensor) -> torch.Tensor:
        """(:911-922) a ground-truth image with an alpha channel over the step's background colour."""
        if image.shape[2] == 4:
            alpha = image[..., -1].unsqueeze(-1).repeat((1, 1, 3))
            return alpha * image[..., :3] + (1 - alpha) * background
        return image

    def get_metrics_dict(self, outputs, batch):
        """(:924-941) without the colour-corrected variant and the camera optimizer (mode "off", :120)."""
        gt_rgb = self.composite_with_background(self.get_gt_img(batch["image"]), outputs["background"])
        mse = torch.nn.functional.mse_loss(outputs["rgb"], gt_rgb)
        metrics = {"psnr": -10.0 * torch.log10(mse)}
        if self.config.color_corrected_metrics:  # (:935-937)
            from .bilagrid import color_correct

            cc = color_correct(outputs["rgb"].detach(), gt_rgb)
            metrics["cc_psnr"] = -10.0 * torch.log10(torch.nn.functional.mse_loss(cc, gt_rgb))
        metrics["gaussian_count"] = self.num_points
        return metrics

    def get_loss_dict(self, outputs, batch, metrics_dict=None):
        """(:944-990) main loss (1 - ssim_lambda) L1 + ssim_lambda (1 - SSIM) on the composited ground truth, the
        optional mask (both images blacked out), the optional scale-ratio regulariser on every 10th step, and while
        training with bilateral grids 10 x their total variation (:988-989).  The camera-optimizer term of the reference
        is not mirrored (mode "off" in every shipped config)."""
        from .harness import l1_and_ssim

        gt_img = self.composite_with_background(self.get_gt_img(batch["image"]), outputs["background"])
        pred_img = outputs["rgb"]
        if "mask" in batch:
            d = self._get_downscale_factor()
            mask = batch["mask"]
            mask = (resize_image(mask.float(), d) if d > 1 else mask.float()).to(self.device)
            assert mask.shape[:2] == gt_img.shape[:2] == pred_img.shape[:2]
            gt_img = gt_img * mask
            pred_img = pred_img * mask
        l1, sim = l1_and_ssim(pred_img, gt_img)  # (fused on the GPU: ops.l1_ssim)
        simloss = 1 - sim
        if self.config.use_scale_regularization and self.step % 10 == 0:
            scale_exp = torch.exp(self.scales)
            ratio = scale_exp.amax(dim=-1) / scale_exp.amin(dim=-1)
            limit = torch.tensor(self.config.max_gauss_ratio, device=ratio.device)
            scale_reg = 0.1 * (torch.maximum(ratio, limit) - self.config.max_gauss_ratio).mean()
        else:
            scale_reg = torch.tensor(0.0, device=self.device)
        lam = self.config.ssim_lambda
        loss_dict = {"main_loss": (1 - lam) * l1 + lam * simloss, "scale_reg": scale_reg}
        if self.training and self.config.use_bilateral_grid:
            from .bilagrid import total_variation_loss

            loss_dict["tv_loss"] = 10 * total_variation_loss(self.bil_grids.grids)
        return loss_dict

    # -- the pieces of get_outputs shared by stage 1 and stage 2 ------------------------------------
    def _camera_setup(self, camera: Camera):
        """viewmat, K, W, H at the scheduled resolution (:806-815)."""
        cam0 = camera.metadata.get("cameras0")
        if cam0 is not None and cam0 is not camera:  # (:802-804)
            assert bool((cam0.get_intrinsics_matrices() == camera.get_intrinsics_matrices()).all()), \
                "Intrinsics matrices should be the same"
        s = self._get_downscale_factor()
        # The reference rescales by 1/s and later by s (:807-808, :813-814).  With floor rounding that
        # does not restore a size that s does not divide (1014 -> 253 -> 1012): its camera objects
        # shrink once and, one schedule step later, render 506 rows against a 507-row target.  Here the
        # cameras get their own sizes back exactly, so render and get_gt_img agree at every step.
        saved = [(c, (c.fx, c.fy, c.cx, c.cy, c.width, c.height)) for c in ((camera, cam0) if cam0 is not None and cam0 is not camera else (camera,))]
        camera.rescale_output_resolution(1 / s)
        if cam0 is not None and cam0 is not camera:  # (:808, :814) rescaled and restored alongside
            cam0.rescale_output_resolution(1 / s)
        c2w = camera.camera_to_worlds
        if c2w.is_cuda or self.device.type != "cuda":
            viewmat = get_viewmat(c2w.to(self.device))
            K = camera.get_intrinsics_matrices().to(self.device)
        else:
            # host-side pose: 25 floats through a pinned buffer, asynchronously.  A pageable
            # `.to(device)` blocks the host until the queue has drained, i.e. until the previous
            # step's backward has finished -- the host could then never run ahead of the GPU.
            # The staging words come from a ring of pinned slots: a slot is written again 64 camera set-ups later, and
            # the host is never that far ahead of the queue (every render waits for its own list length).
            stage, stage_np = _pose_slot()
            stage[:16] = get_viewmat(c2w.float()).reshape(-1)
            stage_np[16:] = (camera.fx, 0.0, camera.cx, 0.0, camera.fy, camera.cy, 0.0, 0.0, 1.0)
            dev = stage.to(self.device, non_blocking=True)
            viewmat, K = dev[:16].view(1, 4, 4), dev[16:].view(1, 3, 3)
        W, H = int(camera.width), int(camera.height)
        _plain_set(self, "last_size", (H, W))
        for c, (fx, fy, cx, cy, w, h) in saved:
            c.fx, c.fy, c.cx, c.cy, c.width, c.height = fx, fy, cx, cy, w, h
        return viewmat, K, W, H

    def _colors_and_degree(self):
        """(:801, :826-830)"""
        colors = torch.cat((self.features_dc[:, None, :], self.features_rest), dim=1)
        if self.config.sh_degree > 0:
            return colors, min(self.step // self.config.sh_degree_interval, self.config.sh_degree)
        return torch.sigmoid(colors), None

    def _render_mode(self) -> str:
        if self.config.rasterize_mode not in ("antialiased", "classic"):
            raise ValueError("Unknown rasterize_mode: %s", self.config.rasterize_mode)
        return "RGB+ED" if (self.config.output_depth_during_training or not self.training) else "RGB"

    def _rasterize_and_finish(self, means, quats, scales, colors, sh_degree, viewmat, K, W, H):
        """The raster call with the reference's exact kwargs (:847-868) and O1 (:869-898)."""
        render_mode = self._render_mode()
        render, alpha, info = rasterization(
            means=means,
            quats=quats,
            scales=scales,
            opacities=torch.sigmoid(self.opacities).squeeze(-1),
            colors=colors,
            viewmats=viewmat,
            Ks=K,
            width=W,
            height=H,
            tile_size=16,
            packed=False,
            near_plane=0.01,
            far_plane=1e10,
            render_mode=render_mode,
            sh_degree=sh_degree,
            sparse_grad=False,
            absgrad=True,
            rasterize_mode=self.config.rasterize_mode,
        )
        if self.training and info["means2d"].requires_grad:
            info["means2d"].retain_grad()
        _plain_set(self, "xys", info["means2d"])  # [1,N,2]
        _plain_set(self, "radii", info["radii"][0])  # [N]
        _plain_set(self, "last_list_length", int(info["raster_flatten_ids"].numel()))  # (the capacity, in static-shape mode)
        background = self._get_background_color()
        rgb = torch.clamp(render[..., :3] + (1 - alpha) * background, 0.0, 1.0)
        if render_mode == "RGB+ED":
            depth = render[..., 3:4]
            depth = torch.where(alpha > 0, depth, depth.detach().max()).squeeze(0)
        else:
            depth = None
        if not self.training:
            background = background.expand(H, W, 3)
        return {"rgb": rgb.squeeze(0), "depth": depth, "accumulation": alpha.squeeze(0), "background": background}

    def _render(self, means, d_rotation, d_scaling, viewmat, K, W, H):
        """Activations (:801, :826-830, :844-851) + raster call + O1, either folded into the HIP
        passes (``fused_front_end``; SH path only) or spelled out in torch exactly like the
        reference."""
        if self.config.fused_front_end and self.config.sh_degree > 0:
            return self._rasterize_raw_and_finish(means, d_rotation, d_scaling, viewmat, K, W, H)
        colors, sh_degree = self._colors_and_degree()
        scales = torch.exp(self.scales) + d_scaling
        quats = self.quats / self.quats.norm(dim=-1, keepdim=True) + d_rotation
        return self._rasterize_and_finish(means, quats, scales, colors, sh_degree, viewmat, K, W, H)

    def _rasterize_raw_and_finish(self, means, d_rotation, d_scaling, viewmat, K, W, H):
        from .rasterization import rasterize_gauss_params

        render_mode = self._render_mode()
        background = self._get_background_color()
        sh_degree = min(self.step // self.config.sh_degree_interval, self.config.sh_degree)
        rgb, alpha, info = rasterize_gauss_params(
            means, self.quats, self.scales, self.opacities, self.features_dc, self.features_rest, viewmat, K, W, H,
            sh_degree,
            d_quats=d_rotation if torch.is_tensor(d_rotation) else None,
            d_scales=d_scaling if torch.is_tensor(d_scaling) else None,
            background=background, clamp=True, near_plane=0.01, far_plane=1e10, tile_size=16,
            render_mode=render_mode, absgrad=True, rasterize_mode=self.config.rasterize_mode,
        )  # fmt: skip
        if self.training and info["means2d"].requires_grad:
            info["means2d"].retain_grad()
        _plain_set(self, "xys", info["means2d"])
        _plain_set(self, "radii", info["radii"][0])
        _plain_set(self, "last_list_length", int(info["raster_flatten_ids"].numel()))  # (the capacity, in static-shape mode)
        if render_mode == "RGB+ED":
            depth = rgb[..., 3:4]
            depth = torch.where(alpha > 0, depth, depth.detach().max()).squeeze(0)
        else:
            depth = None
        if not self.training:
            background = background.expand(H, W, 3)
        # (a full-width slice is still a SliceBackward node: a zeros + a copy launch in every backward)
        return {"rgb": (rgb if rgb.shape[-1] == 3 else rgb[..., :3]).squeeze(0), "depth": depth, "accumulation": alpha.squeeze(0),
                "background": background}  # fmt: skip

    # -- H1 + H4 -----------------------------------------------------------------------------------
    def get_outputs(self, camera: Camera) -> Dict[str, Union[torch.Tensor, List, None]]:
        if not isinstance(camera, Camera):
            print("Called get_outputs with not a camera")
            return {}
        if "cameras0" not in camera.metadata:
            camera.metadata["cameras0"] = camera
        if self.training:
            assert camera.shape[0] == 1, "Only one camera at a time"
        crop = self._crop_ids()
        if crop is not None and int(crop.sum()) == 0:  # (:781-782)
            return self.get_empty_outputs(int(camera.width), int(camera.height), self.background_color.to(self.device))
        _plain_set(self, "_active_crop", crop)
        try:
            return self._get_outputs_on_active_rows(camera)
        finally:
            _plain_set(self, "_active_crop", None)

    def _get_outputs_on_active_rows(self, camera: Camera):
        viewmat, K, W, H = self._camera_setup(camera)
        return self._bilateral(self._outputs_from(viewmat, K, W, H, camera.times), camera)

    def _bilateral(self, out, camera: Camera):
        """(:879-882) while training, the rendered image through the bilateral grid of its training camera."""
        if self.config.use_bilateral_grid and self.training and camera.metadata is not None and "cam_idx" in camera.metadata:
            from .bilagrid import apply_to_render

            rgb = out["rgb"]
            out["rgb"] = apply_to_render(self.bil_grids, rgb.unsqueeze(0), camera.metadata["cam_idx"], rgb.shape[0], rgb.shape[1]).squeeze(0)
        return out

    def _outputs_from(self, viewmat, K, W, H, times):
        """H4 + the raster call on device-resident camera data: everything of ``get_outputs`` behind the host
        side of the camera (what ``graphed.GraphedModelStep`` captures)."""
        if self.step < self.config.warm_up:
            means = self.means
            d_rotation, d_scaling = 0.0, 0.0
        else:
            pts = self.means
            times = times.to(self.device).expand(pts.shape[0], -1)
            d_xyz, d_rotation, d_scaling = self.deform(pts.detach(), times)
            means = transform_points(d_xyz, pts)  # (:840-843; not torch.bmm: utils.small_bmm)
        return self._render(means, d_rotation, d_scaling, viewmat, K, W, H)

    @torch.no_grad()
    def get_outputs_for_camera(self, camera: Camera):
        """(:992-1003)"""
        return self.get_outputs(camera)

    # -- S1 -----------------------------------------------------------------------------------------
    def step_cb(self, step: int) -> None:
        self.step = step

    def after_train_iter(self, step: int) -> None:
        """Densification statistics from ``xys.absgrad`` and ``radii`` (:369-392)."""
        assert step == self.step
        if self.step >= self.config.stop_split_at:
            return
        with torch.no_grad():
            if self.xys_grad_norm is None:
                self.xys_grad_norm = torch.zeros(self.num_points, device=self.device)
                self.vis_counts = torch.ones(self.num_points, device=self.device)
            if self.max_2Dsize is None:
                self.max_2Dsize = torch.zeros(self.num_points, device=self.device)
            absgrad, radii = self.xys.absgrad[0], self.radii.flatten()
            if absgrad.is_cuda:
                # one pass over the Gaussians (csrc/densify.hip) instead of nine torch launches
                from . import ops

                ops.densify_stats(absgrad, radii, float(max(self.last_size)), self.xys_grad_norm, self.vis_counts, self.max_2Dsize)
                return
            # The reference indexes with the boolean mask (`x[visible] += ...`, :379-392): every such line is
            # a nonzero() with a host sync and a gather / scatter pair.  The same numbers with fixed shapes:
            # adding 0.0 and taking max(x, 0) with x >= 0 leave the invisible rows bit-for-bit as they were.
            visible = radii > 0
            grads = absgrad.norm(dim=-1)
            self.vis_counts += visible.to(self.vis_counts.dtype)
            self.xys_grad_norm += torch.where(visible, grads, torch.zeros_like(grads))
            new = radii.float() / float(max(self.last_size))
            self.max_2Dsize = torch.maximum(self.max_2Dsize, torch.where(visible, new, torch.zeros_like(new)))

    def get_gaussian_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        return {k: [self.gauss_params[k]] for k in ("means", "scales", "quats", "features_dc", "features_rest", "opacities")}

    def get_param_groups(self) -> Dict[str, List[nn.Parameter]]:
        groups = self.get_gaussian_param_groups()
        groups["deform"] = list(self.deform.parameters())
        groups["control"] = list(self.control.parameters())
        if self.config.use_bilateral_grid:  # (:617-618)
            groups["bilateral_grid"] = list(self.bil_grids.parameters())
        return groups


class FreeGaussianControlModel(FreeGaussianModel):
    """Stage 2 (reference freegaussian_control_model.py): a frozen deform net supplies the mean
    displacement of each attribute's Gaussians; the control MLP turns it into per-Gaussian
    deltas scattered into the full set; then the same raster call."""

    def __init__(self, gaussian_mask: torch.Tensor, init_camera: Camera, **kw):
        super().__init__(**kw)
        self.register_buffer("gaussian_mask", gaussian_mask.bool())  # [N,M] (gaussian_mask_NxM.npy)
        self.init_camera = copy.deepcopy(init_camera)
        self.step = 30000  # load_state_dict forces this (:280): SH maxed, warm-up over
        self.control_values: Optional[torch.Tensor] = None  # viewer-supplied [M,3] when no cameras0

    def get_outputs(self, camera: Camera):
        if not isinstance(camera, Camera):
            print("Called get_outputs with not a camera")
            return {}
        crop = self._crop_ids()  # (freegaussian_control_model.py:73-86: the mask is cropped alongside)
        if crop is not None and int(crop.sum()) == 0:
            return self.get_empty_outputs(int(camera.width), int(camera.height), self.background_color.to(self.device))
        _plain_set(self, "_active_crop", crop)
        try:
            return self._get_outputs_on_active_rows(camera)
        finally:
            _plain_set(self, "_active_crop", None)

    def _get_outputs_on_active_rows(self, camera: Camera):
        viewmat, K, W, H = self._camera_setup(camera)
        gmask = self.gaussian_mask if self._active_crop is None else self.gaussian_mask[self._active_crop]
        all_means, all_scales, all_quats = self.means, self.scales, self.quats
        sel = gmask.any(-1)
        pts = all_means[sel]
        pmask = gmask[sel]  # [n,M]
        if not self.training and "cameras0" not in camera.metadata and self.control_values is not None:
            d_avg = self.control_values.to(self.device)
        else:
            with torch.no_grad():  # (:128-138)
                def deformed(t):
                    T, _, _ = self.deform(pts, t.to(self.device).expand(pts.shape[0], -1))
                    return transform_points(T, pts)

                delta = deformed(camera.times) - deformed(self.init_camera.times)
                d_avg = torch.stack([delta[pmask[:, i]].mean(0) for i in range(pmask.shape[1])])
        value = pmask.float() @ d_avg / pmask.sum(-1, keepdim=True)  # (:140)
        d_xyz, d_rot, d_scale = self.control(pts, value)
        idx = (sel.nonzero().squeeze(-1),)
        means = all_means + torch.zeros_like(all_means).index_put(idx, d_xyz)
        d_scaling = torch.zeros_like(all_scales).index_put(idx, d_scale)
        d_rotation = torch.zeros_like(all_quats).index_put(idx, d_rot)
        return self._bilateral(self._render(means, d_rotation, d_scaling, viewmat, K, W, H), camera)  # (control model :190-193)

    def get_param_groups(self):
        groups = super().get_param_groups()
        groups.pop("deform")  # (:215-218)
        return groups
