"""Host-side math around the raster call (plain torch, device-agnostic; all tiny next to the
raster).  Mirrors the behaviour of the reference's ``freegaussian/utils.py`` helpers that sit on
the path into ``rasterization`` -- verified against golden vectors captured from the reference
(tests/golden/g_utils.npz) -- re-derived, not transcribed."""
from __future__ import annotations

import math
from typing import Callable, Tuple

import torch
import torch.nn.functional as F

SH_C0 = 0.28209479177387814


def to_homogenous(v: torch.Tensor) -> torch.Tensor:
    """[...,3] -> [...,4] with w = 1 (reference utils.py:59-68)."""
    return F.pad(v, (0, 1), value=1.0)


def from_homogenous(v: torch.Tensor) -> torch.Tensor:
    """[...,4] -> [...,3] / w (reference utils.py:71-80)."""
    return v[..., :3] / v[..., 3:4]


def get_viewmat(camera_to_world: torch.Tensor) -> torch.Tensor:
    """[B,3,4] OpenGL camera-to-world -> [B,4,4] OpenCV world-to-camera (reference
    utils.py:162-179): flip the camera y/z axes, then invert the rigid transform analytically."""
    R = camera_to_world[:, :3, :3] * camera_to_world.new_tensor([1.0, -1.0, -1.0])
    t = camera_to_world[:, :3, 3:]
    Rt = R.transpose(1, 2)
    out = camera_to_world.new_zeros(camera_to_world.shape[0], 4, 4)
    out[:, :3, :3] = Rt
    out[:, :3, 3:] = -(Rt @ t)
    out[:, 3, 3] = 1.0
    return out


def small_bmm(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Batched product of tiny matrices, [N,a,b] x [N,b,c] -> [N,a,c], as ONE broadcast multiply and a sum.
    torch.bmm hands N independent 3x3 / 4x4 products to the BLAS as a strided-batched GEMM: on an MI355X that is 25 ms
    for 300k Gaussians and 86 ms for 1M -- with an equally slow backward -- for work of a few megaflops (the transform
    of the means alone was half of a whole training iteration after warm-up; profiles/r03_train_step.txt)."""
    return (A.unsqueeze(-1) * B.unsqueeze(-3)).sum(-2)


def transform_points(T: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """[N,4,4] rigid / projective transforms applied to [N,3] points, homogeneous divide included: what the reference
    writes as from_homogenous(bmm(T, to_homogenous(pts)[..., None])[..., 0]) (freegaussian_model.py:840-843)."""
    h = (T[:, :, :3] * pts.unsqueeze(-2)).sum(-1) + T[:, :, 3]
    return from_homogenous(h)


def _hat(w: torch.Tensor) -> torch.Tensor:
    """[N,3] -> [N,3,3] cross-product matrices."""
    o = torch.zeros_like(w[:, 0])
    return torch.stack([o, -w[:, 2], w[:, 1], w[:, 2], o, -w[:, 0], -w[:, 1], w[:, 0], o], -1).view(-1, 3, 3)


def exp_se3(screw: torch.Tensor, theta: torch.Tensor) -> torch.Tensor:
    """Screw axis [N,6] = (w, v) and magnitude theta [N,1] -> [N,4,4] rigid transforms
    (Modern Robotics eq. 3.88; behaviour of reference utils.py:137-159)."""
    w, v = screw[:, :3], screw[:, 3:]
    th = theta.reshape(-1, 1, 1)
    W = _hat(w)
    W2 = small_bmm(W, W)
    eye = torch.eye(3, device=screw.device, dtype=screw.dtype).expand_as(W)
    s, c = torch.sin(th), torch.cos(th)
    R = eye + s * W + (1.0 - c) * W2
    G = th * eye + (1.0 - c) * W + (th - s) * W2
    p = small_bmm(G, v.unsqueeze(-1))
    top = torch.cat([R, p], dim=-1)
    # (built on the device: a host list -> device copy is not allowed under hipGraph capture)
    bottom = top.new_zeros(top.shape[0], 1, 4)
    bottom[:, :, 3] = 1.0
    return torch.cat([top, bottom], dim=1)


def positional_encoding(x: torch.Tensor, num_freqs: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] along the last axis."""
    outs = [x]
    for k in range(num_freqs):
        f = float(2**k)
        outs += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(outs, dim=-1)


def get_embedder(multires: int, input_dims: int = 1) -> Tuple[Callable[[torch.Tensor], torch.Tensor], int]:
    """(embed_fn, out_dim), same contract as reference utils.py:8-23."""
    if input_dims == -1:
        return torch.nn.Identity(), 3
    return (lambda x: positional_encoding(x, multires)), input_dims * (1 + 2 * multires)


def RGB2SH(rgb: torch.Tensor) -> torch.Tensor:
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh: torch.Tensor) -> torch.Tensor:
    return sh * SH_C0 + 0.5


def resize_image(image: torch.Tensor, d: int) -> torch.Tensor:
    """[H,W,C] -> [H//d, W//d, C] box ('area') downscale (reference utils.py:248-261)."""
    x = image.to(torch.float32).permute(2, 0, 1).unsqueeze(0)
    return F.avg_pool2d(x, kernel_size=d, stride=d).squeeze(0).permute(1, 2, 0)


def random_quat_tensor(n: int) -> torch.Tensor:
    """Uniform random rotations as [n,4] quaternions (Shoemake; reference utils.py:214-229)."""
    u, v, w = torch.rand(n), torch.rand(n), torch.rand(n)
    a, b = torch.sqrt(1 - u), torch.sqrt(u)
    return torch.stack([a * torch.sin(2 * math.pi * v), a * torch.cos(2 * math.pi * v),
                        b * torch.sin(2 * math.pi * w), b * torch.cos(2 * math.pi * w)], dim=-1)  # fmt: skip


def knn_mean_distance(x: torch.Tensor, k: int = 3) -> torch.Tensor:
    """[N,3] -> [N,1] mean Euclidean distance to the k nearest OTHER points: the initial scale of every Gaussian
    (reference freegaussian_model.py:158-162 through ``k_nearest_sklearn``, :293-311 -- sklearn's NearestNeighbors with
    k + 1 neighbours, the point itself dropped).  The same tree query when sklearn imports; otherwise exact brute force
    in row chunks (the two agree to rounding: both are exact searches).  Fewer than k + 1 points: the neighbours there are."""
    n = x.shape[0]
    if n <= 1:
        return torch.ones(n, 1)
    kk = min(k, n - 1)
    xc = x.detach().cpu().float()
    try:
        from sklearn.neighbors import NearestNeighbors

        d, _ = NearestNeighbors(n_neighbors=kk + 1, algorithm="auto", metric="euclidean").fit(xc.numpy()).kneighbors(xc.numpy())
        return torch.from_numpy(d[:, 1:].astype("float32")).mean(dim=-1, keepdim=True)
    except ImportError:
        out = torch.empty(n, 1)
        for i in range(0, n, 2048):
            d = torch.cdist(xc[i : i + 2048], xc)
            out[i : i + 2048, 0] = d.topk(kk + 1, dim=1, largest=False).values[:, 1:].mean(dim=1)
        return out


def bilinear_interp(image: torch.Tensor, x: torch.Tensor, y: torch.Tensor, reference_quirk: bool = False):
    """Sample [B,H,W,C] at [B,N] pixel coordinates -> [B,N,C].

    ``reference_quirk=True`` reproduces reference utils.py:316-343 exactly: it takes floor/ceil
    corners, so at exactly-integer coordinates all four weights vanish and the result is 0
    (SURVEY.md §8c G5 -- recorded, and available for parity; the default is true bilinear)."""
    B, h, w, _ = image.shape
    b = torch.arange(B, device=image.device)[:, None]
    x0f, y0f = torch.floor(x), torch.floor(y)
    if reference_quirk:
        x1f, y1f = torch.ceil(x), torch.ceil(y)
    else:
        x1f, y1f = x0f + 1, y0f + 1
    x0, x1 = x0f.clamp(0, w - 1).long(), x1f.clamp(0, w - 1).long()
    y0, y1 = y0f.clamp(0, h - 1).long(), y1f.clamp(0, h - 1).long()
    if reference_quirk:
        wx1, wx0 = x - x0, x1 - x
        wy1, wy0 = y - y0, y1 - y
    else:
        wx1, wy1 = x - x0f, y - y0f
        wx0, wy0 = 1 - wx1, 1 - wy1
    return ((wx0 * wy0)[..., None] * image[b, y0, x0] + (wx0 * wy1)[..., None] * image[b, y1, x0]
            + (wx1 * wy0)[..., None] * image[b, y0, x1] + (wx1 * wy1)[..., None] * image[b, y1, x1])  # fmt: skip
