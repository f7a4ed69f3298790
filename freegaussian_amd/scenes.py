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


# ------------------------------------------------------------------------------------------------
# A procedural multi-view TARGET for end-to-end training (scripts/train_e2e.py): not a cloud but SURFACES -- what a
# captured scene is made of and what the reference's data parsers deliver (freegaussian_dataparser.py:317-678: rooms and
# table-top objects, cameras among the content).  A floor, two walls (the other two sides open: some views see past the
# content and carry alpha < 1, like the reference's RGBA synthetic sets), three hollow shells, a table top and thin rods,
# each covered by flat (surfaces) or elongated (rods) Gaussians at a low-discrepancy point set, textured procedurally.


def _quat_from_frame(t1: torch.Tensor, t2: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    """wxyz quaternions of the rotations whose columns are (t1, t2, n) [N,3] each (right-handed, orthonormal)."""
    m00, m10, m20 = t1[:, 0], t1[:, 1], t1[:, 2]
    m01, m11, m21 = t2[:, 0], t2[:, 1], t2[:, 2]
    m02, m12, m22 = n[:, 0], n[:, 1], n[:, 2]
    # the numerically safe branch per row: the largest of (w, x, y, z) squared
    q2 = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], 1).clamp_min(0)
    best = q2.argmax(1)
    w = torch.stack([q2[:, 0], m21 - m12, m02 - m20, m10 - m01], 1)
    x = torch.stack([m21 - m12, q2[:, 1], m01 + m10, m02 + m20], 1)
    y = torch.stack([m02 - m20, m01 + m10, q2[:, 2], m12 + m21], 1)
    z = torch.stack([m10 - m01, m02 + m20, m12 + m21, q2[:, 3]], 1)
    cand = torch.stack([w, x, y, z], 1)  # [N, branch, 4]
    q = cand[torch.arange(len(best)), best]
    return q / q.norm(dim=1, keepdim=True)


def _tangent_frame(n: torch.Tensor, spin: torch.Tensor):
    """Two unit tangents of the unit normals ``n`` [N,3], rotated in the tangent plane by ``spin`` [N] radians."""
    a = torch.where(n[:, 0:1].abs() < 0.9, torch.tensor([1.0, 0.0, 0.0]).expand_as(n), torch.tensor([0.0, 1.0, 0.0]).expand_as(n))
    u = torch.linalg.cross(n, a)
    u = u / u.norm(dim=1, keepdim=True)
    v = torch.linalg.cross(n, u)
    c, s = torch.cos(spin)[:, None], torch.sin(spin)[:, None]
    t1 = c * u + s * v
    return t1, torch.linalg.cross(n, t1)


def _texture(p: torch.Tensor, palette: torch.Tensor, freq: float, checker: float = 0.0) -> torch.Tensor:
    """A smooth procedural albedo in [0.05, 0.95]: three sinusoid mixes of the position blended between the rows of
    ``palette`` [3,3]; ``checker`` > 0 multiplies a soft checkerboard of that period in."""
    a = 0.5 + 0.5 * torch.sin(freq * (1.3 * p[:, 0] + 0.7 * p[:, 1]) + 0.5 * torch.sin(freq * 0.9 * p[:, 2]))
    b = 0.5 + 0.5 * torch.sin(freq * (0.8 * p[:, 2] - 0.6 * p[:, 0]) + 1.7)
    rgb = (1 - a)[:, None] * palette[0] + a[:, None] * ((1 - b)[:, None] * palette[1] + b[:, None] * palette[2])
    if checker > 0:
        k = torch.sin(math.pi * p[:, 0] / checker) * torch.sin(math.pi * p[:, 2] / checker)
        rgb = rgb * (0.75 + 0.25 * torch.tanh(4.0 * k))[:, None]
    return rgb.clamp(0.05, 0.95)


def room_scene(n_gauss: int = 200_000, width: int = 1920, height: int = 1080, n_views: int = 40, sh_degree: int = 3,
               seed: int = 42, focal: Optional[float] = None):  # fmt: skip
    """-> (Scene, meta).  The hidden target model of scripts/train_e2e.py: ``n_gauss`` anisotropic Gaussians laid out as
    surfaces inside [-4,4]^3 (y points up on screen with `look_at_viewmat`'s default, the floor at y = -1.6), ``n_views`` cameras at mixed radii -- two rings inside
    the room, a ring above it looking down, and poses AMONG the objects looking outwards; a fifth of the views is held out
    (every fourth pose of each kind).  ``meta``: ``train`` / ``test`` view indices, ``times`` [V] (four shared instants),
    ``parts`` (name, first, count), ``kinds`` (name, first view, count)."""
    g = torch.Generator().manual_seed(seed)
    sob = torch.quasirandom.SobolEngine(2, scramble=True, seed=seed)
    floor_y, top_y = -1.6, 2.4
    shells = [((-1.2, floor_y + 0.7, 0.5), 0.7), ((1.0, floor_y + 1.0, -0.8), 1.0), ((0.3, floor_y + 0.5, 1.9), 0.5)]
    n_rods = 16
    areas = {"floor": 64.0, "wall_z": 8.0 * (top_y - floor_y), "wall_x": 8.0 * (top_y - floor_y), "table": 1.7 * 2.0}
    for i, (_, r) in enumerate(shells):
        areas[f"shell{i}"] = 4 * math.pi * r * r
    rod_share = 0.015
    total = sum(areas.values())
    counts = {k: int((1 - rod_share) * n_gauss * a / total) for k, a in areas.items()}
    counts["rods"] = n_gauss - sum(counts.values())
    pal = {
        "floor": torch.tensor([[0.75, 0.7, 0.6], [0.35, 0.3, 0.25], [0.85, 0.8, 0.75]]),
        "wall_z": torch.tensor([[0.3, 0.45, 0.7], [0.8, 0.8, 0.85], [0.2, 0.3, 0.5]]),
        "wall_x": torch.tensor([[0.7, 0.35, 0.3], [0.9, 0.8, 0.6], [0.5, 0.2, 0.2]]),
        "table": torch.tensor([[0.45, 0.3, 0.15], [0.6, 0.4, 0.2], [0.3, 0.2, 0.1]]),
        "shell0": torch.tensor([[0.9, 0.2, 0.2], [0.95, 0.7, 0.2], [0.6, 0.1, 0.3]]),
        "shell1": torch.tensor([[0.2, 0.7, 0.3], [0.8, 0.9, 0.3], [0.1, 0.4, 0.4]]),
        "shell2": torch.tensor([[0.3, 0.3, 0.9], [0.7, 0.5, 0.9], [0.2, 0.7, 0.9]]),
    }
    P, Nrm, S, Col, parts = [], [], [], [], []
    first = 0
    for name, cnt in counts.items():
        if name == "rods" or cnt == 0:
            continue
        uv = sob.draw(cnt)
        if name == "floor":
            p = torch.stack([uv[:, 0] * 8 - 4, torch.full((cnt,), floor_y), uv[:, 1] * 8 - 4], 1)
            nrm = torch.tensor([0.0, 1.0, 0.0]).expand(cnt, 3)
            col = _texture(p, pal[name], 1.4, checker=1.0)
        elif name == "wall_z":
            p = torch.stack([uv[:, 0] * 8 - 4, floor_y + uv[:, 1] * (top_y - floor_y), torch.full((cnt,), 4.0)], 1)
            nrm = torch.tensor([0.0, 0.0, -1.0]).expand(cnt, 3)
            col = _texture(p, pal[name], 2.2)
        elif name == "wall_x":
            p = torch.stack([torch.full((cnt,), -4.0), floor_y + uv[:, 1] * (top_y - floor_y), uv[:, 0] * 8 - 4], 1)
            nrm = torch.tensor([1.0, 0.0, 0.0]).expand(cnt, 3)
            col = _texture(p, pal[name], 1.8)
        elif name == "table":
            p = torch.stack([1.5 + uv[:, 0] * 1.7, torch.full((cnt,), -0.6), 1.0 + uv[:, 1] * 2.0], 1)
            nrm = torch.tensor([0.0, 1.0, 0.0]).expand(cnt, 3)
            col = _texture(p, pal[name], 6.0)
        else:
            c, r = shells[int(name[-1])]
            zc = 1 - 2 * uv[:, 0]  # equal-area map of the unit square onto the sphere
            ph = 2 * math.pi * uv[:, 1]
            rr = torch.sqrt((1 - zc * zc).clamp_min(0))
            nrm = torch.stack([rr * torch.cos(ph), zc, rr * torch.sin(ph)], 1)
            p = torch.tensor(c) + r * nrm
            col = _texture(p * 2.5, pal[name], 3.0)
        d = math.sqrt(areas[name] / cnt)  # mean spacing of the point set
        s_in = 0.8 * d * torch.exp(0.25 * torch.randn(cnt, generator=g))
        ratio = torch.exp(0.4 * torch.randn(cnt, generator=g)).clamp(0.4, 2.5)
        sc = torch.stack([s_in * ratio.sqrt(), s_in / ratio.sqrt(), 0.1 * s_in], 1)
        P.append(p), Nrm.append(nrm), S.append(sc), Col.append(col)
        parts.append((name, first, cnt))
        first += cnt
    P, Nrm, S, Col = torch.cat(P), torch.cat(Nrm), torch.cat(S), torch.cat(Col)
    t1, t2 = _tangent_frame(Nrm, torch.rand(P.shape[0], generator=g) * 2 * math.pi)
    Q = _quat_from_frame(t1, t2, Nrm)
    # rods: segments between random points above the floor, Gaussians strung along them, long axis = the rod
    n_rod_pts = counts["rods"]
    if n_rod_pts > 0:
        per = [n_rod_pts // n_rods + (1 if i < n_rod_pts % n_rods else 0) for i in range(n_rods)]
        rp, rq, rs, rc = [], [], [], []
        for i, m in enumerate(per):
            if m == 0:
                continue
            a = torch.tensor([-3.0, floor_y, -3.0]) + torch.rand(3, generator=g) * torch.tensor([6.0, 0.0, 6.0])
            b = a + torch.tensor([0.0, 1.0, 0.0]) * (1.0 + 2.0 * float(torch.rand(1, generator=g))) + (torch.rand(3, generator=g) - 0.5) * torch.tensor([1.6, 0.0, 1.6])
            axis = (b - a) / (b - a).norm()
            L = float((b - a).norm())
            s = (torch.arange(m) + 0.5) / m
            p = a + s[:, None] * (b - a)
            ax = axis.expand(m, 3)
            u1, u2 = _tangent_frame(ax, torch.zeros(m))  # (u1, u2, axis) right-handed with the rod as the third column
            rq.append(_quat_from_frame(u1, u2, ax))
            rs.append(torch.stack([torch.full((m,), 0.02), torch.full((m,), 0.02), torch.full((m,), 0.9 * L / m)], 1))
            hue = torch.rand(3, generator=g) * 0.7 + 0.2
            rc.append((hue * (0.8 + 0.2 * torch.sin(12.0 * s))[:, None]).clamp(0.05, 0.95))
            rp.append(p)
        parts.append(("rods", first, n_rod_pts))
        P, Q, S, Col = torch.cat([P] + rp), torch.cat([Q] + rq), torch.cat([S] + rs), torch.cat([Col] + rc)
        Nrm = torch.cat([Nrm, torch.zeros(n_rod_pts, 3)])
    N = P.shape[0]
    opac = torch.sigmoid(3.0 + torch.randn(N, generator=g))
    K_sh = 16
    colors = torch.zeros(N, K_sh, 3)
    colors[:, 0] = (Col - 0.5) / 0.28209479177387814
    # view dependence: a sheen along the normal in the degree-1 band (y, z, x order) + a little of everything above
    colors[:, 1] = 0.12 * Nrm[:, 1:2]
    colors[:, 2] = 0.12 * Nrm[:, 2:3]
    colors[:, 3] = 0.12 * Nrm[:, 0:1]
    colors[:, 4:] = 0.02 * torch.randn(N, K_sh - 4, 3, generator=g)
    if focal is None:
        focal = 1200.0 * width / 1920.0
    eyes, targets = [], []
    n_inner = n_views // 5  # poses among the objects, looking outwards
    n_top = n_views // 5
    n_mid = n_views // 5
    n_ring = n_views - n_inner - n_top - n_mid
    for i in range(n_ring):
        a = 2 * math.pi * (i + 0.25) / n_ring
        eyes.append([3.3 * math.sin(a), 0.3, -3.3 * math.cos(a)]), targets.append([0.0, -0.7, 0.0])
    for i in range(n_mid):
        a = 2 * math.pi * (i + 0.6) / n_mid
        eyes.append([2.5 * math.sin(a), -0.3, -2.5 * math.cos(a)]), targets.append([0.3 * math.sin(a + 2.0), -0.9, 0.3])
    for i in range(n_top):
        a = 2 * math.pi * (i + 0.1) / n_top
        eyes.append([3.5 * math.sin(a), 2.0, -3.5 * math.cos(a)]), targets.append([0.0, -1.2, 0.0])
    for i in range(n_inner):
        a = 2 * math.pi * (i + 0.4) / n_inner
        r = 0.45 + 0.35 * (i % 2)
        e = [r * math.sin(a), -0.9 + 0.5 * (i % 3) / 2, r * math.cos(a)]
        eyes.append(e), targets.append([e[0] + 3.0 * math.sin(a), -0.9, e[2] + 3.0 * math.cos(a)])
    for i, e in enumerate(eyes):  # no camera inside a closed shell or right at its surface
        inner = i >= n_views - n_inner
        for c, r in shells:
            d = [e[k] - c[k] for k in range(3)]
            dist = math.sqrt(sum(x * x for x in d)) or 1.0
            if dist < r + 0.5:
                for k in range(3):
                    e[k] = c[k] + d[k] / dist * (r + 0.5)
                if inner:  # ... and the poses among the objects look away from the one they stand next to
                    h = math.hypot(d[0], d[2]) or 1.0
                    targets[i] = [e[0] + 3.0 * d[0] / h, -0.9, e[2] + 3.0 * d[2] / h]
    vms = torch.stack([look_at_viewmat(torch.tensor(eyes[i]), torch.tensor(targets[i])) for i in range(n_views)])
    K = torch.tensor([[focal, 0.0, width / 2.0], [0.0, focal, height / 2.0], [0.0, 0.0, 1.0]])
    scene = Scene(P, Q, S, opac, colors, sh_degree, vms, K.expand(n_views, 3, 3).clone(), width, height)
    # held out: every fourth (rings) / eighth (top, inner) pose of each kind's own sequence -- its neighbours either side stay
    # in the training set --, a fifth of the views in all -- the "every k-th frame of the trajectory" protocol
    kinds = [("ring", 0, n_ring), ("mid", n_ring, n_mid), ("top", n_ring + n_mid, n_top), ("inner", n_ring + n_mid + n_top, n_inner)]
    test = []
    for j, (_, first_v, cnt_v) in enumerate(kinds):
        test += [first_v + i for i in range(1 + j, cnt_v, 4 if j < 2 else 8)]  # (ring and mid: every 4th; top and inner: every 8th)
    # time stamps: a multi-view rig (the DyNeRF setting of BASELINE.json) -- four instants, every one seen by a quarter of
    # the cameras of each kind; the target itself does not move
    meta = {"train": [i for i in range(n_views) if i not in test], "test": test,
            "times": [(i % 4) / 4.0 for i in range(n_views)], "parts": parts,
            "kinds": [(k, f, c) for k, f, c in kinds]}  # fmt: skip
    return scene, meta


def load_trained_scene(path: str) -> Scene:
    """The ``trained_scene.npz`` scripts/train_e2e.py writes: the rasterizer's inputs of a TRAINED model (activations and
    deformation applied) with eight of its own cameras -- ``bench.py --layout trained:<path>``."""
    import numpy as np

    z = np.load(path)
    t = {k: torch.from_numpy(np.ascontiguousarray(z[k])).float() for k in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks")}
    return Scene(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], int(z["sh_degree"]), t["viewmats"], t["Ks"],
                 int(z["width"]), int(z["height"]))  # fmt: skip


def scene_statistics(scene: Scene, radii: Optional[torch.Tensor] = None) -> dict:
    """What tells a trained scene from a synthetic cloud: axis-ratio and (given one view's ``radii``) screen-radius
    histograms, the opacity distribution."""
    s = scene.scales.float()
    ratio = s.max(dim=1).values / s.min(dim=1).values.clamp_min(1e-12)
    edges = [1, 1.5, 2, 3, 5, 10, 20, 50, 1e9]
    out = {"axis_ratio_hist": {f"<{edges[i + 1]:g}": int(((ratio >= edges[i]) & (ratio < edges[i + 1])).sum()) for i in range(len(edges) - 1)},
           "axis_ratio_median": float(ratio.median()),
           "opacity_hist": {f"<{b:g}": int(((scene.opacities >= a) & (scene.opacities < b)).sum())
                            for a, b in ((0, 0.02), (0.02, 0.1), (0.1, 0.3), (0.3, 0.7), (0.7, 0.95), (0.95, 1.01))}}  # fmt: skip
    if radii is not None:
        r = radii.flatten().float().cpu()
        r = r[r > 0]
        re = [0, 2, 4, 8, 16, 32, 64, 128, 256, 1e9]
        out["radius_px_hist"] = {f"<{re[i + 1]:g}": int(((r >= re[i]) & (r < re[i + 1])).sum()) for i in range(len(re) - 1)}
        out["radius_px_median"] = float(r.median()) if r.numel() else 0.0
        out["radius_px_max"] = float(r.max()) if r.numel() else 0.0
    return out
