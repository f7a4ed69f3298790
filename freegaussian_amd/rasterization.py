"""Drop-in replacement for ``gsplat.rendering.rasterization`` as the reference calls it.

Call sites this mirrors (kwargs, return values, error behaviour):
  * ``freegaussian/freegaussian_model.py:847-868``          training / eval, ``packed=False``
  * ``freegaussian/freegaussian_control_model.py:158-179``  stage 2, same kwargs
  * ``preprocess/knn_gaussian.py:93-113`` etc.              ``packed=True``, ``render_mode="ED"``

plus the two helpers the reference imports from gsplat: ``quat_to_rotmat``
(``freegaussian_model.py:15,535``) and ``num_sh_bases`` (``:21,165``).

Everything below the argument checks runs in ``libfgraster.so`` (hand-written gfx950 HIP); the
Python here only sequences the stages and owns the autograd tape.  There is no CPU path."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

RENDER_MODES = ("RGB", "D", "ED", "RGB+D", "RGB+ED")


def num_sh_bases(degree: int) -> int:
    """Number of SH basis functions for ``degree`` (reference use: freegaussian_model.py:165)."""
    return (degree + 1) ** 2


def quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """[M,4] wxyz (any norm) -> [M,3,3]; component order of reference utils.py:287-301.
    Used on the host by split_gaussians (freegaussian_model.py:535): plain torch, any device."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )  # fmt: skip
    return R.reshape(quats.shape[:-1] + (3, 3))


class _Info(dict):
    """``info`` dict whose expensive, rarely-read entries are computed on first access."""

    def lazy(self, key, thunk):
        self.__dict__.setdefault("_thunks", {})[key] = thunk

    def __missing__(self, key):
        thunks = self.__dict__.get("_thunks", {})
        if key in thunks:
            self[key] = thunks.pop(key)()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self.__dict__.get("_thunks", {})


def _attach_lists(info: "_Info", tile_keys, flatten_ids, offsets, reference_lists, means2d, radii, depths, tiles,
                  tile_size, tile_w, tile_h) -> None:
    """The list entries of ``info``.  ``raster_flatten_ids`` / ``raster_isect_offsets`` are the lists
    the compositing walked (``last_ids`` indexes them).  ``flatten_ids`` / ``isect_offsets`` /
    ``tile_keys`` / ``isect_ids`` are the REFERENCE's lists (radius-box tile rectangles, sorted by
    tile | depth bits): the same objects when the lists were binned from the radius boxes, otherwise
    rebuilt on first access (footprint rectangles drop entries that contribute to no pixel; nothing
    on the hot path reads the full lists)."""
    info["raster_flatten_ids"], info["raster_isect_offsets"] = flatten_ids, offsets
    if reference_lists:
        info["flatten_ids"], info["isect_offsets"] = flatten_ids, offsets

        # (the thunks must not refer to `info` itself: a reference cycle would keep its tensors -- and
        # the autograd graph behind them -- alive until the cyclic collector runs)
        def _tile_keys():
            return tile_keys if tile_keys is not None else ops.tile_keys_from_offsets(offsets, flatten_ids.numel())

        info.lazy("tile_keys", _tile_keys)
        info.lazy("isect_ids", lambda: ops.isect_keys(_tile_keys(), flatten_ids, depths.detach()))
        return
    cache = {}

    def full():
        if not cache:
            m2 = means2d.detach()
            tk, ids, offs = ops.bin_tiles(m2[0] if m2.dim() == 3 else m2, radii, depths.detach(), tiles, tile_size, tile_w,
                                          tile_h, want_keys=True)  # fmt: skip
            cache.update(tile_keys=tk, flatten_ids=ids, isect_offsets=offs)
        return cache

    info.lazy("flatten_ids", lambda: full()["flatten_ids"])
    info.lazy("isect_offsets", lambda: full()["isect_offsets"])
    info.lazy("tile_keys", lambda: full()["tile_keys"])
    info.lazy("isect_ids", lambda: ops.isect_keys(full()["tile_keys"], full()["flatten_ids"], depths.detach()))


def _empty_result(means, width, height, tile_size, render_mode, sh_degree, colors, extra, packed):
    dev = means.device
    n_rgb = (3 if sh_degree is not None else colors.shape[1]) if render_mode.startswith("RGB") else 0
    C = n_rgb + int(render_mode.endswith("D")) + (0 if extra is None else extra.shape[1])
    tile_w, tile_h = (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size
    zi = torch.zeros(0, dtype=torch.int32, device=dev)
    zf = means.new_zeros(0)
    info = _Info({
        "radii": zi[None] if not packed else zi, "means2d": means.new_zeros((1, 0, 2) if not packed else (0, 2)),
        "depths": zf[None] if not packed else zf, "conics": means.new_zeros((1, 0, 3) if not packed else (0, 3)),
        "opacities": zf[None], "tile_width": tile_w, "tile_height": tile_h, "tiles_per_gauss": zi[None],
        "tile_keys": zi, "flatten_ids": zi, "isect_ids": torch.zeros(0, dtype=torch.int64, device=dev),
        "isect_offsets": torch.zeros(tile_w * tile_h + 1, dtype=torch.int32, device=dev),
        "raster_flatten_ids": zi, "raster_isect_offsets": torch.zeros(tile_w * tile_h + 1, dtype=torch.int32, device=dev),
        "last_ids": torch.full((height, width), -1, dtype=torch.int32, device=dev),
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": 1,
    })  # fmt: skip
    if packed:
        info.update(camera_ids=zi.long(), gaussian_ids=zi.long())
    return means.new_zeros(1, height, width, C), means.new_zeros(1, height, width, 1), info


def _refuse_camera_gradients(viewmats: torch.Tensor, Ks: torch.Tensor, fused: bool = True) -> None:
    """What the raster path does NOT differentiate, said out loud instead of handed back as a silent zero.  The camera
    POSE is differentiable on the fused path (``fg_viewmat_bwd``: the reference keeps a ``CameraOptimizer`` in the loop,
    freegaussian_model.py:120 -- mode "off" in every shipped config --, applied at :774; with another mode ``viewmats``
    requires a gradient); the intrinsics never are, and the stage-by-stage operators (``fused=False``) have no pose
    gradient either."""
    if not torch.is_grad_enabled():
        return
    if Ks.requires_grad:
        raise NotImplementedError("Ks requires a gradient: the raster path does not differentiate the intrinsics; detach() them")
    if viewmats.requires_grad and not fused:
        raise NotImplementedError(
            "viewmats requires a gradient: camera-pose gradients (camera_optimizer mode != 'off', "
            "freegaussian_model.py:120,774) come from the fused path only; call with fused=True or detach() the camera")


def rasterization(
    means: torch.Tensor,  # [N,3]
    quats: torch.Tensor,  # [N,4] wxyz, normalised by the callee
    scales: torch.Tensor,  # [N,3]
    opacities: torch.Tensor,  # [N]
    colors: torch.Tensor,  # [N,K,3] SH coefficients, or [N,C] when sh_degree is None
    viewmats: torch.Tensor,  # [1,4,4] world->camera
    Ks: torch.Tensor,  # [1,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[torch.Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    extra_channels: Optional[torch.Tensor] = None,
    fused: bool = True,
    ctx: Optional["ops.RasterContext"] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """Render one camera.  Returns ``(render [1,H,W,C], alpha [1,H,W,1], info)``.
    ``ctx`` (an extension): the ``ops.RasterContext`` -- launch policy, list-capacity history, hooks -- this
    call and its backward use; default: the calling thread's current one (``ops.use``), else the process's.

    ``info["means2d"]`` is a non-leaf tensor in the graph ([1,N,2], or [nnz,2] when packed):
    ``.retain_grad()`` works on it and after backward it carries ``.absgrad`` when
    ``absgrad=True`` (reference freegaussian_model.py:869-872, :377).  ``info["radii"]`` is
    int32 [1,N] (``>0`` <=> visible).  ``extra_channels`` [N,E] (an extension) are composited
    like colours and appended after the render-mode channels -- used for the flow channels.
    ``fused=False`` runs the stage-by-stage operators (one C-ABI call per stage of SURVEY §8a)
    instead of the fused per-Gaussian passes; both give the same results."""
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if render_mode not in RENDER_MODES:
        raise ValueError(f"Unknown render_mode: {render_mode}")
    if sparse_grad:
        raise NotImplementedError("sparse_grad=True is never used by the reference (freegaussian_model.py:863)")
    if viewmats.dim() != 3 or viewmats.shape[0] != 1 or Ks.shape[0] != 1:
        raise ValueError("exactly one camera per call (reference asserts camera.shape[0]==1)")
    _refuse_camera_gradients(viewmats, Ks, fused and not ops.current().overlap_pack)
    N = means.shape[0]
    if not (means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3) and opacities.shape == (N,)):
        raise ValueError("means[N,3] quats[N,4] scales[N,3] opacities[N] expected")
    if sh_degree is None:
        if colors.dim() != 2 or colors.shape[0] != N:
            raise ValueError("colors must be [N,C] when sh_degree is None")
    else:
        if colors.dim() != 3 or colors.shape[0] != N or colors.shape[2] != 3:
            raise ValueError("colors must be [N,K,3] SH coefficients when sh_degree is given")
        if (sh_degree + 1) ** 2 > colors.shape[1]:
            raise ValueError("sh_degree too large for the given coefficients")

    if not means.is_cuda:
        from ._lib import FgRasterError

        raise FgRasterError("rasterization needs CUDA/HIP tensors: the raster path has no CPU fallback")
    if N == 0:  # nothing to draw: empty image, empty lists (no kernel is launched on empty buffers)
        return _empty_result(means, width, height, tile_size, render_mode, sh_degree, colors, extra_channels, packed)
    if ctx is not None:
        with ops.use(ctx):
            return rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane,
                                 far_plane, radius_clip, eps2d, sh_degree, packed, tile_size, backgrounds, render_mode,
                                 sparse_grad, absgrad, rasterize_mode, extra_channels, fused)  # fmt: skip

    viewmat = viewmats[0]
    K = Ks[0]
    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    with_rgb, with_depth = render_mode.startswith("RGB"), render_mode.endswith("D")
    antialiased = rasterize_mode == "antialiased"
    n_extra = 0 if extra_channels is None else extra_channels.shape[1]

    step = None
    if fused:
        # one pass: projection + SH colour + 64-byte record per Gaussian
        n_color = (3 if sh_degree is not None else colors.shape[1]) if with_rgb else 0
        channels = n_color + int(with_depth) + n_extra
        if not 1 <= channels <= ops.MAX_CHANNELS:
            raise ValueError(f"1..{ops.MAX_CHANNELS} composited channels supported, got {channels}")
        if ops.step_path_available(ops.current(), N, width, height, tile_size, means.device):
            # the whole view as one C-ABI call per direction (fg_step_fwd / fg_step_bwd): same kernels, same results
            step = ops.raster_step(means, quats, scales, opacities, colors if with_rgb else None, viewmat, K, width, height,
                                   extra=extra_channels, eps2d=eps2d, near_plane=near_plane, far_plane=far_plane,
                                   radius_clip=radius_clip, antialiased=antialiased,
                                   sh_degree=(sh_degree if (sh_degree is not None and with_rgb) else -1),
                                   with_depth=with_depth, absgrad=absgrad)  # fmt: skip
    if step is not None:
        (render, alpha, means2d_n, depths, conics, last_ids, radii, tiles, splats, flatten_ids, offsets, node) = step
        opac, tile_keys, keys_rects = splats[:, 2], None, True
    elif fused:
        radii, means2d_n, depths, conics, tiles, splats = ops.preprocess(
            means, quats, scales, opacities, colors if with_rgb else None, extra_channels, viewmat, K, width,
            height, eps2d, near_plane, far_plane, radius_clip, tile_size, antialiased,
            (sh_degree if (sh_degree is not None and with_rgb) else -1), with_depth, overlap=True,
        )  # fmt: skip
        opac = splats[:, 2]
    else:
        radii, means2d_n, depths, conics, comp, tiles = ops.project(
            means, quats, scales, viewmat, K, width, height, eps2d, near_plane, far_plane, radius_clip, tile_size,
            calc_compensations=antialiased,
        )  # fmt: skip
        opac = opacities.float()
        if antialiased:
            opac = opac * comp
        if sh_degree is None:
            rgb = colors.float()
        else:
            rgb = ops.spherical_harmonics(sh_degree, means, viewmat, colors, radii)
        chans = []
        if with_rgb:
            chans.append(rgb)
        if with_depth:
            chans.append(depths[:, None])
        if extra_channels is not None:
            chans.append(extra_channels.float())
        feats = chans[0] if len(chans) == 1 else torch.cat(chans, dim=-1)

    if step is None:
        # speculative lists: the raster forward is enqueued before the host waits for the list length
        keys_rects = getattr(splats, "_fg_bin", None) if fused else None
        tile_keys, flatten_ids, offsets, finish_lists = ops.bin_tiles(
            means2d_n.detach(), radii, depths.detach(), tiles, tile_size, tile_w, tile_h, defer=True, want_keys=False,
            keys_rects=keys_rects, raster_hint=(channels, width, height) if fused else None,
        )

    if packed:
        # the reference's packed call sites (knn_gaussian.py:116-130) index these by nnz
        gids = torch.nonzero(radii > 0).squeeze(-1)
        means2d_info = means2d_n[gids]
        means2d_in = means2d_n
    else:
        means2d_info = means2d_n.unsqueeze(0)  # [1,N,2]: the tensor that is retain_grad()'ed
        means2d_in = means2d_info

    def composite():
        if fused:
            return ops.rasterize_splats(splats, means2d_in, channels, width, height, tile_size, offsets, flatten_ids,
                                        absgrad=absgrad)  # fmt: skip
        return ops.rasterize_to_pixels(means2d_in, conics, feats, opac, width, height, tile_size, offsets, flatten_ids,
                                       absgrad=absgrad)  # fmt: skip

    if step is None:
        render, alpha, last_ids = composite()
        if finish_lists is not None:
            tile_keys, flatten_ids, redone = finish_lists()
            if redone:  # the capacity guess was too small: lists were rebuilt exactly, composite again
                render, alpha, last_ids = composite()
    elif node is not None:
        import weakref

        node.means2d_ref = weakref.ref(means2d_in)  # the tensor that receives .grad / .absgrad after backward
    if backgrounds is not None:
        render = render + (1.0 - alpha) * backgrounds.reshape(1, 1, -1)
    if render_mode in ("ED", "RGB+ED"):
        di = ((3 if sh_degree is not None else colors.shape[1]) if with_rgb else 0)  # depth follows the colours
        d = render[..., di : di + 1] / alpha.clamp(min=1e-10)
        render = torch.cat([render[..., :di], d, render[..., di + 1 :]], dim=-1)

    info = _Info({
        "radii": radii[None],
        "means2d": means2d_info,
        "depths": depths[None],
        "conics": conics[None],
        "opacities": opac[None],
        "tile_width": tile_w,
        "tile_height": tile_h,
        "tiles_per_gauss": tiles[None],
        "last_ids": last_ids,
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": 1,
    })
    _attach_lists(info, tile_keys, flatten_ids, offsets, keys_rects is None, means2d_n, radii, depths, tiles, tile_size,
                  tile_w, tile_h)  # fmt: skip
    if packed:
        info.update(
            camera_ids=torch.zeros_like(gids),
            gaussian_ids=gids,
            radii=radii[gids],
            depths=depths[gids],
            conics=conics[gids],
        )
    return render[None], alpha[None], info


def rasterize_gauss_params(
    means: torch.Tensor,  # [N,3] (already deformed, if a deform net is active)
    quats: torch.Tensor,  # [N,4] raw parameter, normalised in-kernel
    log_scales: torch.Tensor,  # [N,3] raw parameter, exp() in-kernel
    opacity_logits: torch.Tensor,  # [N,1] or [N] raw parameter, sigmoid() in-kernel
    features_dc: torch.Tensor,  # [N,3]
    features_rest: torch.Tensor,  # [N,K-1,3]
    viewmats: torch.Tensor,
    Ks: torch.Tensor,
    width: int,
    height: int,
    sh_degree: int,
    d_quats: Optional[torch.Tensor] = None,  # [N,4] added after normalisation (:845)
    d_scales: Optional[torch.Tensor] = None,  # [N,3] added after exp (:844)
    background: Optional[torch.Tensor] = None,  # [3]; composited in the raster epilogue
    clamp: bool = False,  # clamp the colour channels to [0,1] (:876)
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    tile_size: int = 16,
    render_mode: str = "RGB",
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    extra_channels: Optional[torch.Tensor] = None,
    ctx: Optional["ops.RasterContext"] = None,
) -> Tuple[torch.Tensor, torch.Tensor, Dict]:
    """The model-side front end of SURVEY.md section 8f row 3: what FreeGaussianModel.get_outputs
    does around the raster call (freegaussian_model.py:801 SH ``cat``, :844-851 ``exp`` /
    ``sigmoid`` / quaternion normalisation + MLP deltas, :875-877 background composite + clamp)
    folded into the two per-Gaussian passes and the raster epilogue/prologue.  Same outputs as
    ``rasterization(means, quats/|quats| + d_quats, exp(log_scales) + d_scales,
    sigmoid(opacity_logits), cat(dc, rest), ...)`` followed by the composite, unpacked layout;
    the returned ``render`` is the finished image when ``background``/``clamp`` are given
    (depth, if any, stays accumulated-then-normalised as in ``rasterization``)."""
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if render_mode not in RENDER_MODES:
        raise ValueError(f"Unknown render_mode: {render_mode}")
    if viewmats.dim() != 3 or viewmats.shape[0] != 1 or Ks.shape[0] != 1:
        raise ValueError("exactly one camera per call (reference asserts camera.shape[0]==1)")
    _refuse_camera_gradients(viewmats, Ks)
    if not means.is_cuda:
        from ._lib import FgRasterError

        raise FgRasterError("rasterize_gauss_params needs CUDA/HIP tensors: the raster path has no CPU fallback")
    if not render_mode.startswith("RGB"):
        raise ValueError("the raw-parameter path renders colour: use rasterization() for depth-only modes")
    if ctx is not None:
        with ops.use(ctx):
            return rasterize_gauss_params(means, quats, log_scales, opacity_logits, features_dc, features_rest, viewmats,
                                          Ks, width, height, sh_degree, d_quats, d_scales, background, clamp, near_plane,
                                          far_plane, radius_clip, eps2d, tile_size, render_mode, absgrad, rasterize_mode,
                                          extra_channels)  # fmt: skip
    N = means.shape[0]
    if N == 0:  # nothing to draw (everything culled away): the background, empty lists
        render, alpha, info = _empty_result(means, width, height, tile_size, render_mode, sh_degree, None,
                                            extra_channels, False)  # fmt: skip
        if background is not None:
            render = render.clone()
            render[..., :3] = background.detach().reshape(1, 1, 1, 3).to(render)
        if clamp:
            render = torch.cat([render[..., :3].clamp(0.0, 1.0), render[..., 3:]], dim=-1)
        return render, alpha, info
    with_depth = render_mode.endswith("D")
    n_extra = 0 if extra_channels is None else extra_channels.shape[1]
    channels = 3 + int(with_depth) + n_extra
    if channels > ops.MAX_CHANNELS:
        raise ValueError(f"1..{ops.MAX_CHANNELS} composited channels supported, got {channels}")
    viewmat, K = viewmats[0], Ks[0]
    tile_w = (width + tile_size - 1) // tile_size
    tile_h = (height + tile_size - 1) // tile_size
    bg = None
    if background is not None:  # (RGB: the tensor itself; more channels: one pad op -- no zeros + slice copy)
        bg = background.detach().reshape(-1).to(means.device, torch.float32)
        if channels > 3:
            bg = torch.nn.functional.pad(bg, (0, channels - 3))
    if ops.step_path_available(ops.current(), N, width, height, tile_size, means.device):
        # the whole view as one C-ABI call per direction (fg_step_fwd / fg_step_bwd): same kernels, same results
        import weakref

        (render, alpha, means2d_info, depths, conics, last_ids, radii, tiles, splats, flatten_ids, offsets, node) = ops.raster_step(
            means, quats, log_scales, opacity_logits, features_dc, viewmat, K, width, height, raw=True, d_quats=d_quats,
            d_scales=d_scales, features_rest=features_rest, extra=extra_channels, background=bg, n_clamp=(3 if clamp else 0),
            eps2d=eps2d, near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
            antialiased=(rasterize_mode == "antialiased"), sh_degree=sh_degree, with_depth=with_depth, absgrad=absgrad,
            batched=True)  # fmt: skip
        if node is not None:
            node.means2d_ref = weakref.ref(means2d_info)
        if with_depth:
            d = render[..., 3:4] / alpha.clamp(min=1e-10)
            render = torch.cat([render[..., :3], d, render[..., 4:]], dim=-1)
        info = _Info({
            "radii": radii[None], "means2d": means2d_info, "depths": depths[None], "conics": conics[None],
            "opacities": splats[:, 2][None], "tile_width": tile_w, "tile_height": tile_h,
            "tiles_per_gauss": tiles[None], "last_ids": last_ids, "width": width, "height": height,
            "tile_size": tile_size, "n_cameras": 1,
        })  # fmt: skip
        _attach_lists(info, None, flatten_ids, offsets, False, means2d_info, radii, depths, tiles, tile_size, tile_w, tile_h)
        return render, alpha, info
    radii, means2d_n, depths, conics, tiles, splats = ops.preprocess_raw(
        means, quats, log_scales, opacity_logits, features_dc, features_rest, viewmat, K, width, height, sh_degree,
        d_quats=d_quats, d_scales=d_scales, extra=extra_channels, eps2d=eps2d, near_plane=near_plane,
        far_plane=far_plane, radius_clip=radius_clip, tile_size=tile_size,
        antialiased=(rasterize_mode == "antialiased"), with_depth=with_depth,
    )  # fmt: skip
    keys_rects = getattr(splats, "_fg_bin", None)
    tile_keys, flatten_ids, offsets, finish_lists = ops.bin_tiles(
        means2d_n.detach(), radii, depths.detach(), tiles, tile_size, tile_w, tile_h, defer=True, want_keys=False,
        keys_rects=keys_rects, raster_hint=(channels, width, height),
    )
    means2d_info = means2d_n.unsqueeze(0)

    def composite():
        return ops.rasterize_splats(splats, means2d_info, channels, width, height, tile_size, offsets, flatten_ids,
                                    absgrad=absgrad, background=bg, n_clamp=(3 if clamp else 0))  # fmt: skip

    render, alpha, last_ids = composite()
    if finish_lists is not None:
        tile_keys, flatten_ids, redone = finish_lists()
        if redone:
            render, alpha, last_ids = composite()
    if with_depth:
        d = render[..., 3:4] / alpha.clamp(min=1e-10)
        render = torch.cat([render[..., :3], d, render[..., 4:]], dim=-1)
    info = _Info({
        "radii": radii[None], "means2d": means2d_info, "depths": depths[None], "conics": conics[None],
        "opacities": splats[:, 2][None], "tile_width": tile_w, "tile_height": tile_h,
        "tiles_per_gauss": tiles[None], "last_ids": last_ids, "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": 1,
    })  # fmt: skip
    _attach_lists(info, tile_keys, flatten_ids, offsets, keys_rects is None, means2d_n, radii, depths, tiles, tile_size,
                  tile_w, tile_h)  # fmt: skip
    return render[None], alpha[None], info
