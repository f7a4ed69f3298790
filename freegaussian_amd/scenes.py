"""Deterministic synthetic scenes for the BASELINE.json configurations.

Generated on the CPU with a fixed seed (42, the seed of the reference YAMLs,
config/sim/base.yaml:8) so the oracle and the HIP path see identical inputs."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Scene:
    means: torch.Tensor  # [N,3]
    quats: torch.Tensor  # [N,4] wxyz, unnormalised
    scales: torch.Tensor  # [N,3] linear (already exp'd)
    opacities: torch.Tensor  # [N] in (0,1)
    colors: torch.Tensor  # [N,16,3] SH coefficients
    sh_degree: int
    viewmats: torch.Tensor  # [V,4,4] world->camera (OpenCV axes)
    Ks: torch.Tensor  # [V,3,3]
    width: int
    height: int

    def to(self, device) -> "Scene":
        kw = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()}
        return Scene(**kw)


def look_at_viewmat(eye: torch.Tensor, target: torch.Tensor, up=(0.0, -1.0, 0.0)) -> torch.Tensor:
    """World->camera with OpenCV axes (x right, y down, z forward)."""
    f = target - eye
    f = f / f.norm()
    upv = torch.tensor(up, dtype=eye.dtype)
    r = torch.linalg.cross(f, -upv)
    if r.norm() < 1e-6:
        r = torch.tensor([1.0, 0.0, 0.0], dtype=eye.dtype)
    r = r / r.norm()
    d = torch.linalg.cross(f, r)
    R = torch.stack([r, d, f], 0)  # rows: camera axes in world coordinates
    vm = torch.eye(4, dtype=eye.dtype)
    vm[:3, :3] = R
    vm[:3, 3] = -R @ eye
    return vm


def _ring_cameras(n_views, radius, fx, fy, cx, cy):
    vms, Ks = [], []
    for v in range(n_views):
        a = 2 * math.pi * v / max(n_views, 1)
        eye = torch.tensor([radius * math.sin(a), 0.0, -radius * math.cos(a)])
        vms.append(look_at_viewmat(eye, torch.zeros(3)))
        Ks.append(torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]))
    return torch.stack(vms), torch.stack(Ks)


def plumbing_scene(seed: int = 42) -> Scene:
    """cfg1: 1k Gaussians, 128x128, one view, SH degree 0."""
    g = torch.Generator().manual_seed(seed)
    N = 1000
    means = torch.rand(N, 3, generator=g) * 2 - 1
    scales = torch.exp(torch.empty(N, 3).uniform_(math.log(0.02), math.log(0.1), generator=g))
    quats = torch.randn(N, 4, generator=g)
    opac = torch.sigmoid(torch.randn(N, generator=g))
    dc = torch.rand(N, 1, 3, generator=g)
    colors = torch.cat([dc, torch.zeros(N, 15, 3)], 1)
    vm = torch.eye(4)
    vm[2, 3] = 3.0  # camera at (0,0,-3) looking down +z
    K = torch.tensor([[128.0, 0.0, 64.0], [0.0, 128.0, 64.0], [0.0, 0.0, 1.0]])
    return Scene(means, quats, scales, opac, colors, 0, vm[None], K[None], 128, 128)


def synthetic_scene(n_gauss: int, width: int, height: int, n_views: int = 1, sh_degree: int = 3, seed: int = 42,
                    focal: Optional[float] = None, extent: float = 2.0, cam_radius: float = 4.0,
                    log_scale_mean: float = math.log(0.01), log_scale_std: float = 0.5) -> Scene:  # fmt: skip
    """cfg2..cfg5 family: means ~ U([-extent,extent]^3); log-scales ~
    N(log 0.01, 0.5^2) clipped to [log 0.002, log 0.05]; quats ~ normalised N(0,1)^4; opacity =
    sigmoid(N(0,1.5^2)); SH dc ~ N(0,1), rest ~ N(0,0.1^2); cameras on a ring of radius 4."""
    g = torch.Generator().manual_seed(seed)
    N = n_gauss
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * extent
    ls = torch.randn(N, 3, generator=g) * log_scale_std + log_scale_mean
    scales = torch.exp(ls.clamp(math.log(0.002), math.log(0.05)))
    quats = torch.randn(N, 4, generator=g)
    opac = torch.sigmoid(torch.randn(N, generator=g) * 1.5)
    dc = torch.randn(N, 1, 3, generator=g)
    rest = torch.randn(N, 15, 3, generator=g) * 0.1
    colors = torch.cat([dc, rest], 1)
    if focal is None:
        focal = 1200.0 * width / 1920.0
    vms, Ks = _ring_cameras(n_views, cam_radius, focal, focal, width / 2.0, height / 2.0)
    return Scene(means, quats, scales, opac, colors, sh_degree, vms, Ks, width, height)


def north_star_scene(n_views: int = 8, seed: int = 42) -> Scene:
    """cfg4: 1M Gaussians, 1920x1080, fx=fy=1200, 8 views 45 degrees apart."""
    return synthetic_scene(1_000_000, 1920, 1080, n_views=n_views, sh_degree=3, seed=seed)


def apply_layout(scene: Scene, layout: str, seed: int = 7) -> Scene:
    """Reshape a ``synthetic_scene`` in place into one of the non-uniform test distributions; parts joined by ``+``:

    * ``uniform`` -- as generated (U([-2,2]^3), near-isotropic Gaussians: three independent log-normal axes);
    * ``clustered:<frac>:<extent>`` -- the first ``frac`` of the Gaussians pulled into a ball of that extent at the centre
      (tile lists of tens of thousands of entries next to empty ones);
    * ``needles:<frac>:<ratio>`` -- a random ``frac`` of the Gaussians made anisotropic the way densification leaves a
      trained scene (reference freegaussian_model.py:524-571: splits shrink all axes alike, so an elongated parent breeds
      elongated children): one axis x ``ratio``, one / 3, the third kept -- needles and plates at random orientations (the
      quaternions are already uniform on the sphere), clipped to 0.3 scene units."""
    N = scene.means.shape[0]
    for part in layout.split("+"):
        if part in ("", "uniform"):
            continue
        f = part.split(":")
        if f[0] == "clustered" and len(f) == 3:
            scene.means[: int(float(f[1]) * N)] *= float(f[2]) / 2.0
        elif f[0] == "needles" and len(f) == 3:
            g = torch.Generator().manual_seed(seed)
            pick = torch.rand(N, generator=g) < float(f[1])
            axes = torch.argsort(torch.rand(N, 3, generator=g), dim=1)  # a random (long, short, kept) assignment per Gaussian
            factor = torch.ones(N, 3)
            factor.scatter_(1, axes[:, 0:1], float(f[2]))
            factor.scatter_(1, axes[:, 1:2], 1.0 / 3.0)
            scene.scales[pick] = (scene.scales[pick] * factor[pick]).clamp(max=0.3)
        else:
            raise ValueError(f"layout {part!r}: uniform | clustered:<frac>:<extent> | needles:<frac>:<ratio> (joined by +)")
    return scene
