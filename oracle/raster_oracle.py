"""CPU oracle for the Gaussian-raster hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``freegaussian_amd``) never does: it fails loudly when
the HIP extension is missing.

PARITY UNPINNED.  The arithmetic of this path does not live in the reference tree: it is the
third-party ``gsplat.rendering.rasterization`` (``pyproject.toml:9``, ``gsplat >= 1.0.0``, no
lockfile, not vendored, not installed, CUDA-only), called at
``freegaussian/freegaussian_model.py:847-868`` and
``freegaussian/freegaussian_control_model.py:158-179``.  The reference has no tests, golden
images or fixtures for this boundary.  This file therefore restates the
*published* gsplat-1.x / 3DGS algorithm (EWA projection, 0.3 px screen-space blur, real SH
basis, 16x16 tile binning with 64-bit ``tile|depth`` keys, stable sort, front-to-back alpha
compositing with the 1/255 skip, 0.999 alpha clamp and 1e-4 transmittance stop), anchored on
the constants the reference's call sites do state:

* tile size 16                     ``freegaussian_model.py:806``
* near 0.01 / far 1e10             ``freegaussian_model.py:859-860``
* "classic" | "antialiased"        ``freegaussian_model.py:818``, blur 0.3 ``:110-119``
* SH C0 = 0.28209479177387814      ``freegaussian/utils.py:236,244``
* wxyz quaternion order            ``freegaussian/utils.py:287-290``
* render modes RGB / RGB+ED / ED   ``freegaussian_model.py:821-824``, ``preprocess/knn_gaussian.py:108``

BACKWARD.  One semantics everywhere: the compositing stage is an autograd node (``_CompositeRef``) whose
backward is the analytic restatement ``rasterize_backward(..., alpha_out=...)`` in the REFERENCE's order --
gsplat's hand-written backward starts each pixel from ``T_final = 1 - alpha_out`` (a value rounded at
ulp(1): 6e-4 relative on a saturated pixel) and rebuilds every T_i from it by multiplying with
``1 / (1 - alpha_i)`` on the way back; every T-dependent term carries that rounding.  ``fg_oracle.c`` and
the HIP kernels do the same, so all three are held to the same 1e-4 bar.  ``rasterization(...,
backward="autograd")`` keeps the plain autograd of the forward (exact running products T_i) as the
cross-check of the algebra (fp64: 1e-10 against the analytic form with exact T); on deep lists the two
semantics differ by ~1e-4 relative L2 in fp32 (1M Gaussians: 1.4e-4), which is why parity is NOT judged
against the autograd form.

Everything is plain PyTorch on CPU.  The projection is written component-by-component with
a fixed operation order and only IEEE-exact operations (+ - * / sqrt), so that a GPU kernel
compiled without FMA contraction reproduces ``radii``, ``means2d`` and ``depths`` bit for
bit; tile ids and sort keys derived from them are then bit-exact integers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

ALPHA_SKIP = 1.0 / 255.0  # a splat contributing less than this to a pixel is skipped
ALPHA_MAX = 0.999  # alpha clamp
T_STOP = 1e-4  # stop compositing once transmittance would fall to this
EPS2D = 0.3  # screen-space blur added to the 2D covariance diagonal
FOV_CLAMP = 1.3  # Jacobian evaluated inside 1.3x the field of view

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (
    -0.5900435899266435,
    2.890611442640554,
    -0.4570457994644658,
    0.3731763325901154,
    -0.4570457994644658,
    1.445305721320277,
    -0.5900435899266435,
)


def num_sh_bases(degree: int) -> int:
    """(degree+1)^2 -- replacement contract for ``gsplat.cuda_legacy._wrapper.num_sh_bases``
    used at ``freegaussian_model.py:165``."""
    return (degree + 1) ** 2


def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """wxyz quaternion (normalised here) -> [M,3,3].  Contract of
    ``gsplat.cuda_legacy._torch_impl.quat_to_rotmat`` used at ``freegaussian_model.py:535``;
    component order per ``freegaussian/utils.py:287-301``."""
    w, x, y, z = quats.unbind(-1)
    n = _sqrt(((w * w + x * x) + y * y) + z * z)
    w, x, y, z = w / n, x / n, y / n, z / n
    R = _rot_components(w, x, y, z)
    return torch.stack(R, -1).reshape(quats.shape[:-1] + (3, 3))


def _sqrt(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded square root.  torch.sqrt on CPU float32 goes through a vectorised
    approximation that is NOT always correctly rounded (measured: 0.7% of inputs off by 1 ulp);
    the double-precision root rounded once to float is (53 >= 2*24+2 bits)."""
    if x.dtype == torch.float32:
        return torch.sqrt(x.double()).float()
    return torch.sqrt(x)


def _rot_components(w, x, y, z):
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    wx, wy, wz = w * x, w * y, w * z
    return (
        1.0 - 2.0 * (y2 + z2), 2.0 * (xy - wz), 2.0 * (xz + wy),
        2.0 * (xy + wz), 1.0 - 2.0 * (x2 + z2), 2.0 * (yz - wx),
        2.0 * (xz - wy), 2.0 * (yz + wx), 1.0 - 2.0 * (x2 + y2),
    )  # fmt: skip


@dataclass
class Projected:
    radii: torch.Tensor  # [N] int32, >0 <=> visible
    means2d: torch.Tensor  # [N,2]
    depths: torch.Tensor  # [N]
    conics: torch.Tensor  # [N,3]  (a, b, c) of the inverse 2D covariance
    compensations: torch.Tensor  # [N] sqrt(det_orig/det_blurred)


def _project_core(means, quats, scales, viewmat, K, width, height, eps2d):
    """Floating-point part of the projection for a set of Gaussians (no culling)."""
    mx, my, mz = means.unbind(-1)
    qw, qx, qy, qz = quats.unbind(-1)
    s0, s1, s2 = scales.unbind(-1)
    W = [[viewmat[i, j] for j in range(3)] for i in range(3)]
    t = [viewmat[i, 3] for i in range(3)]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]

    # camera-space mean
    mc = [((W[i][0] * mx + W[i][1] * my) + W[i][2] * mz) + t[i] for i in range(3)]
    x, y, z = mc

    # 3D covariance  C = (R S)(R S)^T
    qn = _sqrt(((qw * qw + qx * qx) + qy * qy) + qz * qz)
    R = _rot_components(qw / qn, qx / qn, qy / qn, qz / qn)
    M = [[R[3 * i + 0] * s0, R[3 * i + 1] * s1, R[3 * i + 2] * s2] for i in range(3)]

    def dot3(a, b):
        return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]

    C = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            C[i][j] = dot3(M[i], M[j])
            C[j][i] = C[i][j]

    # camera-space covariance  CC = W C W^T
    T = [[(W[i][0] * C[0][j] + W[i][1] * C[1][j]) + W[i][2] * C[2][j] for j in range(3)] for i in range(3)]
    CC = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            CC[i][j] = (T[i][0] * W[j][0] + T[i][1] * W[j][1]) + T[i][2] * W[j][2]
            CC[j][i] = CC[i][j]

    # perspective Jacobian, evaluated at the FOV-clamped point
    # NB: `python_float / tensor` is reciprocal-then-multiply in torch; use a true IEEE division
    tan_fovx = torch.full_like(fx, 0.5 * width) / fx
    tan_fovy = torch.full_like(fy, 0.5 * height) / fy
    lim_x = FOV_CLAMP * tan_fovx
    lim_y = FOV_CLAMP * tan_fovy
    rz = 1.0 / z
    rz2 = rz * rz
    tx = z * torch.minimum(lim_x, torch.maximum(-lim_x, x * rz))
    ty = z * torch.minimum(lim_y, torch.maximum(-lim_y, y * rz))
    ja = fx * rz
    jb = -(fx * tx) * rz2
    jc = fy * rz
    jd = -(fy * ty) * rz2
    u0 = ja * CC[0][0] + jb * CC[0][2]
    u1 = ja * CC[0][1] + jb * CC[1][2]
    u2 = ja * CC[0][2] + jb * CC[2][2]
    w1 = jc * CC[1][1] + jd * CC[1][2]
    w2 = jc * CC[1][2] + jd * CC[2][2]
    c00 = u0 * ja + u2 * jb
    c01 = u1 * jc + u2 * jd
    c11 = w1 * jc + w2 * jd
    m2x = (fx * x) * rz + cx
    m2y = (fy * y) * rz + cy

    # screen-space blur + compensation
    det_orig = c00 * c11 - c01 * c01
    c00 = c00 + eps2d
    c11 = c11 + eps2d
    det = c00 * c11 - c01 * c01
    comp = _sqrt(torch.clamp_min(det_orig / det, 0.0))
    inv_det = 1.0 / det
    conic_a = c11 * inv_det
    conic_b = -c01 * inv_det
    conic_c = c00 * inv_det

    # extent: 3 sigma of the major axis
    b = 0.5 * (c00 + c11)
    v1 = b + _sqrt(torch.clamp_min(b * b - det, 0.01))
    radius_f = torch.ceil(3.0 * _sqrt(v1))
    return m2x, m2y, z, conic_a, conic_b, conic_c, comp, det, radius_f


def project(
    means: torch.Tensor,
    quats: torch.Tensor,
    scales: torch.Tensor,
    viewmat: torch.Tensor,
    K: torch.Tensor,
    width: int,
    height: int,
    eps2d: float = EPS2D,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
) -> Projected:
    """K1: world -> camera -> screen EWA projection with culling.

    Fixed evaluation order; see module docstring.  Differentiable through autograd for every
    floating output; culled Gaussians produce zeros with zero gradient (the differentiable
    pass only ever sees the survivors, so no NaN can leak through a masked branch)."""
    N = means.shape[0]
    dt = means.dtype
    with torch.no_grad():
        m2x, m2y, z, _, _, _, _, det, radius_f = _project_core(
            means, quats, scales, viewmat, K, width, height, eps2d
        )
        ok = (z >= near_plane) & (z <= far_plane) & (det > 0)
        ok &= torch.isfinite(radius_f) & (radius_f > radius_clip)
        ok &= ~((m2x + radius_f <= 0) | (m2x - radius_f >= width) | (m2y + radius_f <= 0) | (m2y - radius_f >= height))
        radii = torch.where(ok, radius_f, torch.zeros_like(radius_f)).to(torch.int32)
        sel = torch.nonzero(ok).squeeze(-1)
    m2x, m2y, z, ca, cb, cc, comp, _, _ = _project_core(
        means[sel], quats[sel], scales[sel], viewmat, K, width, height, eps2d
    )

    def scatter(cols):
        v = torch.stack(cols, -1)
        return torch.zeros(N, v.shape[-1], dtype=dt).index_put((sel,), v)

    return Projected(
        radii=radii,
        means2d=scatter([m2x, m2y]),
        depths=scatter([z])[:, 0],
        conics=scatter([ca, cb, cc]),
        compensations=scatter([comp])[:, 0],
    )


def sh_eval(degree: int, dirs: torch.Tensor, coeffs: torch.Tensor) -> torch.Tensor:
    """K2: real spherical harmonics, 3DGS basis/sign convention.  ``dirs`` [N,3] unnormalised,
    ``coeffs`` [N,K,3] with K >= (degree+1)^2.  Returns [N,3] *without* the +0.5 offset."""
    n = torch.sqrt(((dirs[:, 0] * dirs[:, 0] + dirs[:, 1] * dirs[:, 1]) + dirs[:, 2] * dirs[:, 2]))
    x, y, z = (dirs[:, 0] / n)[:, None], (dirs[:, 1] / n)[:, None], (dirs[:, 2] / n)[:, None]
    sh = coeffs
    res = SH_C0 * sh[:, 0]
    if degree > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if degree > 1:
        xx, yy, zz = x * x, y * y, z * z
        xy, yz, xz = x * y, y * z, x * z
        res = (
            res
            + SH_C2[0] * xy * sh[:, 4]
            + SH_C2[1] * yz * sh[:, 5]
            + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
            + SH_C2[3] * xz * sh[:, 7]
            + SH_C2[4] * (xx - yy) * sh[:, 8]
        )
    if degree > 2:
        res = (
            res
            + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
            + SH_C3[1] * xy * z * sh[:, 10]
            + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
            + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
            + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
            + SH_C3[5] * z * (xx - yy) * sh[:, 14]
            + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15]
        )
    return res


def tile_rects(means2d: torch.Tensor, radii: torch.Tensor, tile_size: int, tile_w: int, tile_h: int):
    """Tile rectangle [min, max) covered by each visible Gaussian; zeros for culled ones."""
    ts = float(tile_size)
    r = radii.to(means2d.dtype) / ts
    tx = means2d[:, 0] / ts
    ty = means2d[:, 1] / ts
    vis = radii > 0

    def lo(v, m):
        return torch.clamp(torch.floor(v).to(torch.int64), 0, m)

    def hi(v, m):
        return torch.clamp(torch.ceil(v).to(torch.int64), 0, m)

    x0, x1 = lo(tx - r, tile_w), hi(tx + r, tile_w)
    y0, y1 = lo(ty - r, tile_h), hi(ty + r, tile_h)
    z = torch.zeros_like(x0)
    return (torch.where(vis, x0, z), torch.where(vis, y0, z), torch.where(vis, x1, z), torch.where(vis, y1, z))


def isect_tiles(means2d, radii, depths, tile_size: int, tile_w: int, tile_h: int, sort: bool = True):
    """K3 + K4: emit one (key, gaussian id) per (Gaussian, overlapped tile), then stable sort.

    key = tile_id << 32 | float32 bits of depth (depth > 0, so the integer order is the float
    order).  Emission order is Gaussian-major, then row-major over the tile rectangle; with a
    stable sort, ties on (tile, depth) therefore stay in ascending Gaussian id.
    Returns (tiles_per_gauss [N] int32, isect_ids [I] int64, flatten_ids [I] int32)."""
    with torch.no_grad():
        x0, y0, x1, y1 = tile_rects(means2d.detach(), radii, tile_size, tile_w, tile_h)
        nx, ny = x1 - x0, y1 - y0
        cnt = nx * ny
        total = int(cnt.sum())
        gid = torch.repeat_interleave(torch.arange(cnt.numel()), cnt)
        start = torch.cumsum(cnt, 0) - cnt
        k = torch.arange(total) - start[gid]
        nxg = torch.clamp_min(nx[gid], 1)
        ty = y0[gid] + k // nxg
        tx = x0[gid] + k % nxg
        tile_id = ty * tile_w + tx
        dbits = depths.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        keys = (tile_id << 32) | dbits[gid]
        vals = gid.to(torch.int32)
        if sort:
            keys, order = torch.sort(keys, stable=True)
            vals = vals[order]
        return cnt.to(torch.int32), keys, vals


def isect_offsets(isect_ids: torch.Tensor, n_tiles: int) -> torch.Tensor:
    """Per-tile start offset into the sorted list, [n_tiles+1] int32 (last = I)."""
    tile_of = (isect_ids >> 32).contiguous()
    bounds = torch.searchsorted(tile_of, torch.arange(n_tiles + 1, dtype=torch.int64))
    return bounds.to(torch.int32)


def _tile_terms(px, py, xy, conic, opac):
    """sigma, raw alpha, validity for [P] pixels x [L] splats."""
    dx = xy[None, :, 0] - px[:, None]
    dy = xy[None, :, 1] - py[:, None]
    sigma = 0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) + conic[None, :, 1] * dx * dy
    alpha = torch.clamp_max(opac[None, :] * torch.exp(-sigma), ALPHA_MAX)
    valid = (sigma >= 0) & (alpha >= ALPHA_SKIP)
    return dx, dy, sigma, alpha, valid


def _composite_state(alpha, valid):
    """Sequential front-to-back transmittance with the T_STOP rule, vectorised.
    Returns (a, T_excl, include, T_final, n_included_index)."""
    a = torch.where(valid, alpha, torch.zeros_like(alpha))
    T_incl = torch.cumprod(1.0 - a, dim=1)  # sequential product: same order as the kernel
    stop = valid & (T_incl <= T_STOP)
    stopped = torch.cumsum(stop.to(torch.int32), dim=1) > 0  # this one and all later are dropped
    include = valid & ~stopped
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    return a, T_excl, include, stopped


def rasterize(
    means2d: torch.Tensor,
    conics: torch.Tensor,
    colors: torch.Tensor,
    opacities: torch.Tensor,
    width: int,
    height: int,
    tile_size: int,
    offsets: torch.Tensor,
    flatten_ids: torch.Tensor,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """K5: per-tile front-to-back compositing.  colors [N,C].  Returns render [H,W,C],
    alpha [H,W,1], last_ids [H,W] int32 (index into the sorted list of the last splat that
    contributed to the pixel; the tile's start offset - 1 if none)."""
    C = colors.shape[1]
    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    render = torch.zeros(height, width, C, dtype=colors.dtype)
    alpha_img = torch.zeros(height, width, 1, dtype=colors.dtype)
    last_ids = torch.zeros(height, width, dtype=torch.int32)
    rows, cols = [], []
    for t in range(tile_w * tile_h):
        s, e = int(offsets[t]), int(offsets[t + 1])
        ty, tx = divmod(t, tile_w)
        y0, x0 = ty * tile_size, tx * tile_size
        y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
        if e <= s:
            last_ids[y0:y1, x0:x1] = s - 1
            continue
        ids = flatten_ids[s:e].long()
        yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        px = xx.reshape(-1).to(colors.dtype) + 0.5
        py = yy.reshape(-1).to(colors.dtype) + 0.5
        _, _, _, alpha, valid = _tile_terms(px, py, means2d[ids], conics[ids], opacities[ids])
        a, T_excl, include, stopped = _composite_state(alpha, valid)
        wgt = torch.where(include, a * T_excl, torch.zeros_like(a))  # [P,L]
        out = wgt @ colors[ids]  # [P,C]
        T_final = torch.where(include, 1.0 - a, torch.ones_like(a)).prod(dim=1)
        with torch.no_grad():
            L = e - s
            idx = torch.arange(L)[None, :].expand_as(include)
            last = torch.where(include, idx, torch.full_like(idx, -1)).max(dim=1).values + s
        render[y0:y1, x0:x1] = out.reshape(y1 - y0, x1 - x0, C)
        alpha_img[y0:y1, x0:x1, 0] = (1.0 - T_final).reshape(y1 - y0, x1 - x0)
        last_ids[y0:y1, x0:x1] = last.reshape(y1 - y0, x1 - x0).to(torch.int32)
    return render, alpha_img, last_ids


def rasterize_backward(
    means2d, conics, colors, opacities, width, height, tile_size, offsets, flatten_ids, v_render, v_alpha,
    alpha_out=None,
):
    """K6 restated analytically (not via autograd) so that ``absgrad`` -- the sum over pixels
    of |dL/d means2d| that ``freegaussian_model.py:377`` reads from ``means2d.absgrad`` -- is
    defined.  Returns (v_means2d, v_means2d_abs, v_conics, v_colors, v_opacities).

    ``alpha_out`` [H,W] or [H,W,1] = the forward's alpha image: the REFERENCE's backward semantics (the
    autograd of the call at ``freegaussian_model.py:847-868``): each pixel starts from
    ``T_final = 1 - alpha_out`` and T_i is rebuilt walking back, ``T *= 1 / (1 - alpha_i)``, exactly the
    order of ``fg_oracle.c::fgo_raster_bwd``.  Without it the exact forward products are used (the
    algebra's own cross-check)."""
    if alpha_out is not None:
        alpha_out = alpha_out.reshape(height, width)
    N, C = colors.shape
    dt = colors.dtype
    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    v_xy = torch.zeros(N, 2, dtype=dt)
    v_xy_abs = torch.zeros(N, 2, dtype=dt)
    v_conic = torch.zeros(N, 3, dtype=dt)
    v_col = torch.zeros(N, C, dtype=dt)
    v_op = torch.zeros(N, dtype=dt)
    for t in range(tile_w * tile_h):
        s, e = int(offsets[t]), int(offsets[t + 1])
        if e <= s:
            continue
        ty, tx = divmod(t, tile_w)
        y0, x0 = ty * tile_size, tx * tile_size
        y1, x1 = min(y0 + tile_size, height), min(x0 + tile_size, width)
        ids = flatten_ids[s:e].long()
        yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
        px = xx.reshape(-1).to(dt) + 0.5
        py = yy.reshape(-1).to(dt) + 0.5
        xy, con, op, col = means2d[ids], conics[ids], opacities[ids], colors[ids]
        dx, dy, sigma, alpha, valid = _tile_terms(px, py, xy, con, op)
        a, T_excl, include, stopped = _composite_state(alpha, valid)
        inc = include.to(dt)
        if alpha_out is None:
            T_final = torch.where(include, 1.0 - a, torch.ones_like(a)).prod(dim=1)  # [P]
        else:
            # reference order: T = 1 - alpha_out;  for i = last .. first:  T *= 1 / (1 - alpha_i)
            T_final = 1.0 - alpha_out[y0:y1, x0:x1].reshape(-1).to(dt)
            r_inc = torch.where(include, 1.0 / (1.0 - a), torch.ones_like(a))
            seq = torch.cat([T_final[:, None], torch.flip(r_inc, [1])], 1)
            T_excl = torch.flip(torch.cumprod(seq, 1)[:, 1:], [1])  # T in front of entry i
        fac = a * T_excl * inc  # [P,L]
        vr = v_render[y0:y1, x0:x1].reshape(-1, C)  # [P,C]
        va = v_alpha[y0:y1, x0:x1].reshape(-1)  # [P]
        # colour gradient
        g_col = fac.t() @ vr  # [L,C]
        # suffix sums S_i = sum_{j>i} c_j fac_j, projected on v_render
        contrib = fac * (col @ vr.t()).t()  # [P,L]  = fac_j * <c_j, v_render>
        suffix = torch.flip(torch.cumsum(torch.flip(contrib, [1]), 1), [1]) - contrib
        ra = 1.0 / (1.0 - a)
        g_alpha = (T_excl * (col @ vr.t()).t() - suffix * ra) + (T_final * va)[:, None] * ra
        g_alpha = g_alpha * inc
        vis = torch.exp(-sigma)
        unclamped = ((op[None, :] * vis) <= ALPHA_MAX).to(dt) * inc
        g_sigma = -(op[None, :] * vis) * g_alpha * unclamped
        g_op = (vis * g_alpha * unclamped).sum(0)
        g_ca = (0.5 * g_sigma * dx * dx).sum(0)
        g_cb = (g_sigma * dx * dy).sum(0)
        g_cc = (0.5 * g_sigma * dy * dy).sum(0)
        gx = g_sigma * (con[None, :, 0] * dx + con[None, :, 1] * dy)
        gy = g_sigma * (con[None, :, 1] * dx + con[None, :, 2] * dy)
        v_xy.index_add_(0, ids, torch.stack([gx.sum(0), gy.sum(0)], -1))
        v_xy_abs.index_add_(0, ids, torch.stack([gx.abs().sum(0), gy.abs().sum(0)], -1))
        v_conic.index_add_(0, ids, torch.stack([g_ca, g_cb, g_cc], -1))
        v_col.index_add_(0, ids, g_col)
        v_op.index_add_(0, ids, g_op)
    return v_xy, v_xy_abs, v_conic, v_col, v_op


class _AbsgradTap(torch.autograd.Function):
    """Identity on (render, alpha) whose backward sees both upstream gradients together and
    evaluates the analytic K6 restatement once more to define ``means2d.absgrad`` (the gradients
    themselves keep flowing through autograd untouched)."""

    @staticmethod
    def forward(ctx, render, alpha, state):
        ctx.state = state
        return render.view_as(render), alpha.view_as(alpha)

    @staticmethod
    def backward(ctx, v_render, v_alpha):
        st = ctx.state
        with torch.no_grad():
            _, v_abs, _, _, _ = rasterize_backward(*st["args"], v_render, v_alpha[..., 0])
        st["holder"].absgrad = v_abs[None]
        return v_render, v_alpha, None


class _CompositeRef(torch.autograd.Function):
    """K5 + K6 as ONE autograd node with the reference's backward semantics (``rasterize_backward`` with
    ``alpha_out``): what ``rasterization`` uses by default.  Also defines ``means2d.absgrad``."""

    @staticmethod
    def forward(ctx, means2d, conics, feats, opac, geom, offsets, flatten_ids, holder):
        width, height, tile_size = geom
        with torch.no_grad():
            render, alpha, last_ids = rasterize(means2d, conics, feats, opac, width, height, tile_size, offsets,
                                                flatten_ids)  # fmt: skip
        ctx.save_for_backward(means2d, conics, feats, opac, offsets, flatten_ids, alpha)
        ctx.geom, ctx.holder = geom, holder
        ctx.mark_non_differentiable(last_ids)
        return render, alpha, last_ids

    @staticmethod
    def backward(ctx, v_render, v_alpha, _v_last):
        means2d, conics, feats, opac, offsets, flatten_ids, alpha = ctx.saved_tensors
        width, height, tile_size = ctx.geom
        if v_render is None:
            v_render = torch.zeros(height, width, feats.shape[1], dtype=feats.dtype)
        if v_alpha is None:
            v_alpha = torch.zeros(height, width, 1, dtype=feats.dtype)
        with torch.no_grad():
            v_xy, v_abs, v_conic, v_col, v_op = rasterize_backward(
                means2d, conics, feats, opac, width, height, tile_size, offsets, flatten_ids, v_render,
                v_alpha[..., 0], alpha_out=alpha)  # fmt: skip
        if ctx.holder is not None:
            ctx.holder.absgrad = v_abs[None]
        return v_xy, v_conic, v_col, v_op, None, None, None, None


@dataclass
class RasterResult:
    render: torch.Tensor  # [1,H,W,C]
    alpha: torch.Tensor  # [1,H,W,1]
    info: dict


def rasterization(
    means,
    quats,
    scales,
    opacities,
    colors,
    viewmats,
    Ks,
    width: int,
    height: int,
    tile_size: int = 16,
    packed: bool = False,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    render_mode: str = "RGB",
    sh_degree: Optional[int] = None,
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    radius_clip: float = 0.0,
    eps2d: float = EPS2D,
    extra_channels: Optional[torch.Tensor] = None,
    compositor=None,
    backward: str = "reference",
):
    """The whole K0 boundary on CPU for one camera, differentiable by autograd.

    ``extra_channels`` [N,E] are composited like colours and appended after the render-mode
    channels (used for the flow channels F1).  ``compositor``: another implementation of the K5/K6
    stage, ``f(means2d, conics, feats, opac, W, H, tile, offsets, flatten_ids, absgrad_holder) ->
    (render, alpha, last_ids)`` -- ``c_oracle.composite`` plugs the scalar C restatement in, which
    makes full-resolution oracle runs affordable (the torch compositing costs ~10 ms per tile).
    ``backward``: "reference" (default; T rebuilt from ``1 - alpha_out``, see the module header) or
    "autograd" (plain autograd of the forward, the exact-T cross-check; only without ``compositor``)."""
    if backward not in ("reference", "autograd"):
        raise ValueError(f"Unknown backward: {backward}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(f"Unknown render_mode: {render_mode}")
    assert viewmats.shape[0] == 1 and Ks.shape[0] == 1, "one camera at a time"
    viewmat, K = viewmats[0], Ks[0]
    N = means.shape[0]
    proj = project(means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip)
    opac = opacities
    if rasterize_mode == "antialiased":
        opac = opacities * proj.compensations
    visible = proj.radii > 0
    if sh_degree is not None:
        campos = torch.linalg.inv(viewmat)[:3, 3]
        dirs = means - campos[None, :]
        # only visible Gaussians are shaded; culled ones get colour 0
        safe_dirs = torch.where(visible[:, None], dirs, torch.ones_like(dirs))
        rgb = torch.clamp_min(sh_eval(sh_degree, safe_dirs, colors) + 0.5, 0.0)
        rgb = torch.where(visible[:, None], rgb, torch.zeros_like(rgb))
    else:
        rgb = colors
    chans = []
    if render_mode.startswith("RGB"):
        chans.append(rgb)
    if render_mode.endswith("D"):
        chans.append(proj.depths[:, None])
    if extra_channels is not None:
        chans.append(extra_channels)
    feats = torch.cat(chans, -1)

    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    # the [1,N,2] tensor handed out in `info` is the one the compositing consumes, so that
    # `info["means2d"].retain_grad()` sees the screen-space gradient (freegaussian_model.py:869-870)
    means2d_out = proj.means2d[None]
    m2 = means2d_out[0]
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(
        proj.means2d, proj.radii, proj.depths, tile_size, tile_w, tile_h
    )
    offsets = isect_offsets(isect_ids, tile_w * tile_h)
    if compositor is not None:
        render, alpha, last_ids = compositor(
            m2, proj.conics, feats, opac, width, height, tile_size, offsets, flatten_ids, means2d_out if absgrad else None
        )
    elif backward == "reference":
        render, alpha, last_ids = _CompositeRef.apply(
            m2, proj.conics, feats, opac, (int(width), int(height), int(tile_size)), offsets, flatten_ids,
            means2d_out if absgrad else None)  # fmt: skip
    else:
        render, alpha, last_ids = rasterize(
            m2, proj.conics, feats, opac, width, height, tile_size, offsets, flatten_ids
        )
    if compositor is None and backward == "autograd" and absgrad and torch.is_grad_enabled() and render.requires_grad:
        args = tuple(t.detach() for t in (proj.means2d, proj.conics, feats, opac)) + (
            width, height, tile_size, offsets, flatten_ids)  # fmt: skip
        render, alpha = _AbsgradTap.apply(render, alpha, {"args": args, "holder": means2d_out})
    if render_mode in ("ED", "RGB+ED"):
        di = 3 if render_mode == "RGB+ED" else 0
        d = render[..., di : di + 1] / alpha.clamp(min=1e-10)
        render = torch.cat([render[..., :di], d, render[..., di + 1 :]], -1)
    info = {
        "radii": proj.radii[None],
        "means2d": means2d_out,
        "depths": proj.depths[None],
        "conics": proj.conics[None],
        "opacities": opac[None],
        "tile_width": tile_w,
        "tile_height": tile_h,
        "tiles_per_gauss": tiles_per_gauss[None],
        "isect_ids": isect_ids,
        "flatten_ids": flatten_ids,
        "isect_offsets": offsets,
        "last_ids": last_ids,
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": 1,
    }
    if packed:
        gids = torch.nonzero(visible).squeeze(-1)
        info.update(
            camera_ids=torch.zeros_like(gids),
            gaussian_ids=gids,
            radii=proj.radii[gids],
            means2d=proj.means2d[gids],
            depths=proj.depths[gids],
            conics=proj.conics[gids],
        )
    return render[None], alpha[None], info


# ---------------------------------------------------------------------------------------
# Flow-derivative restatement (F-spec / F1 / F2)


def camera_flow_AB(x: torch.Tensor, y: torch.Tensor, fx, fy, cx, cy):
    """A (2x3) and B (2x3) at pixel coordinates (x, y), the code's sign convention:
    ``preprocess/epipolar_flow.py:274-305`` (which is -1x ``docs/index.html:266-273``)."""
    one, zero = torch.ones_like(x), torch.zeros_like(x)
    A = torch.stack([one * fx, zero, cx - x, zero, one * fy, cy - y], -1).reshape(x.shape + (2, 3))
    B = torch.stack(
        [
            -(x - cx) * (y - cy) / fy,
            fx + (x - cx) ** 2 / fx,
            -(y - cy) * fx / fy,
            -fy - (y - cy) ** 2 / fy,
            (x - cx) * (y - cy) / fx,
            (x - cx) * fy / fx,
        ],
        -1,
    ).reshape(x.shape + (2, 3))
    return A, B


def camera_flow(Z: torch.Tensor, fx, fy, cx, cy, veloc: torch.Tensor, omega: torch.Tensor) -> torch.Tensor:
    """Per-pixel camera flow ``A v / Z + B w`` (``preprocess/epipolar_flow.py:309``), pixel
    centres at integer coordinates (``get_image_coords(pixel_offset=0)``, ``:272``); infinite
    depth -> 0 (``:315-317``).  Z [H,W]; returns [H,W,2]."""
    H, Wd = Z.shape
    yy, xx = torch.meshgrid(torch.arange(H, dtype=Z.dtype), torch.arange(Wd, dtype=Z.dtype), indexing="ij")
    A, B = camera_flow_AB(xx, yy, fx, fy, cx, cy)
    flow = (A @ veloc.to(Z.dtype)) / Z[..., None] + B @ omega.to(Z.dtype)
    return torch.where(torch.isinf(Z)[..., None], torch.zeros_like(flow), flow)


def gaussian_flow(means2d, depths, vel_cam, fx, fy, cx, cy, veloc, omega):
    """F2: per-Gaussian projection-flow Jacobian terms (Lemma 1, ``docs/index.html:256-273``,
    code sign convention).  For Gaussian i at screen position mu_i with camera depth Z_i and
    camera-frame velocity v_i:  u_gs_i = A(mu_i) v_i / Z_i,  u_cam_i = A(mu_i) v / Z_i + B(mu_i) w.
    Returns (u_gs [N,2], u_cam [N,2])."""
    A, B = camera_flow_AB(means2d[:, 0], means2d[:, 1], fx, fy, cx, cy)
    iz = (1.0 / depths)[:, None]
    u_gs = torch.einsum("nij,nj->ni", A, vel_cam) * iz
    u_cam = torch.einsum("nij,j->ni", A, veloc.to(A.dtype)) * iz + torch.einsum("nij,j->ni", B, omega.to(A.dtype))
    return u_gs, u_cam


def camera_flow_reprojection(Z, Z1, c2w0, c2w1, K, opticalflow=None):
    """F-spec': the exact-reprojection variant, ``preprocess/epipolar_flow_bp.py:258-298``, restated
    line by line INCLUDING its conventions: poses go OpenGL->OpenCV by the camera-axis flip only
    (``manual2cv(..., keep_original_world_coordinate=True)``, ``:229-241``); a pixel is lifted with the
    depth of frame 0 (``:271-273``), mapped by ``inverse(c2w0)`` and then by ``c2w1`` (``:275-276`` --
    i.e. by c2w1 . c2w0^-1, the camera-to-world matrices used where world-to-camera ones would give
    the physical reprojection), divided by the depth map of frame 1 (``:276``) and compared with the
    pixel: ``uv - xy`` (``:279``).  Returned as the reference returns it: ``sceneflow = -(uv - xy)``
    (``:295``), ``interflow = opticalflow - (uv - xy)`` (``:282``), both 0 at infinite depth.
    Z, Z1 [H,W,1]; c2w [3,4]; K [3,3]; pixel centres at integer coordinates (``:268``)."""
    dt = Z.dtype

    def to_cv(c2w):
        m = torch.eye(4, dtype=dt)
        m[:3] = c2w.to(dt)
        m[0:3, 1:3] = -m[0:3, 1:3]
        return m

    m0, m1 = to_cv(c2w0), to_cv(c2w1)
    H, W = Z.shape[:2]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    pix = torch.stack([xx, yy, torch.ones_like(xx)], -1).unsqueeze(-1)  # [H,W,3,1]
    Kd = K.to(dt)
    p_cam = torch.linalg.inv(Kd) @ pix * Z.unsqueeze(-1)
    p_h = torch.cat([p_cam, torch.ones_like(p_cam[..., :1, :])], -2)
    p3d = torch.linalg.inv(m0) @ p_h
    uvf = 1 / Z1.unsqueeze(-1) * (Kd @ (m1 @ p3d)[..., :3, :])
    raw = uvf[..., :2, 0] - torch.stack([xx, yy], -1)  # uv - xy
    inf = torch.isinf(Z).squeeze(-1)
    out = {"sceneflow": torch.where(inf[..., None], torch.zeros_like(raw), -raw), "raw": raw}
    if opticalflow is not None:
        inter = opticalflow.to(dt) - raw
        out["interflow"] = torch.where(inf[..., None], torch.zeros_like(inter), inter)
    return out
