"""ctypes wrapper of oracle/libfg_oracle.so (the plain-C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libfg_oracle.so")
        if not os.path.exists(path):
            import subprocess

            subprocess.check_call(["make", "-C", _HERE])
        _lib = ctypes.CDLL(path)
        _lib.fgo_count_isects.restype = ctypes.c_int64
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _c(t, dtype=torch.float32):
    return t.detach().to(dtype).contiguous()


def project(means, quats, scales, viewmat, K, width, height, eps2d=0.3, near=0.01, far=1e10, radius_clip=0.0):
    N = means.shape[0]
    means, quats, scales, viewmat, K = map(_c, (means, quats, scales, viewmat, K))
    radii = torch.empty(N, dtype=torch.int32)
    m2, d, con, comp = torch.empty(N, 2), torch.empty(N), torch.empty(N, 3), torch.empty(N)
    f = ctypes.c_float
    lib().fgo_project(N, _p(means), _p(quats), _p(scales), _p(viewmat), _p(K), width, height, f(eps2d), f(near), f(far),
                      f(radius_clip), _p(radii), _p(m2), _p(d), _p(con), _p(comp))  # fmt: skip
    return radii, m2, d, con, comp


def isect_tiles(means2d, radii, depths, ts, tw, th, sort=True):
    N = radii.shape[0]
    means2d, depths, radii = _c(means2d), _c(depths), _c(radii, torch.int32)
    cnt = torch.empty(N, dtype=torch.int32)
    n = lib().fgo_count_isects(N, _p(means2d), _p(radii), ts, tw, th, _p(cnt))
    keys, vals = torch.empty(n, dtype=torch.int64), torch.empty(n, dtype=torch.int32)
    lib().fgo_isect_sorted(N, _p(means2d), _p(radii), _p(depths), ts, tw, th, ctypes.c_int64(n), _p(keys), _p(vals),
                           int(sort))  # fmt: skip
    return cnt, keys, vals


def tile_offsets(keys, n_tiles):
    offs = torch.empty(n_tiles + 1, dtype=torch.int32)
    lib().fgo_tile_offsets(ctypes.c_int64(keys.numel()), _p(keys), n_tiles, _p(offs))
    return offs


def raster_fwd(means2d, conics, feats, opac, width, height, ts, offsets, ids):
    C = feats.shape[1]
    means2d, conics, feats, opac = map(_c, (means2d, conics, feats, opac))
    render, alphas = torch.empty(height, width, C), torch.empty(height, width, 1)
    last = torch.empty(height, width, dtype=torch.int32)
    lib().fgo_raster_fwd(C, width, height, ts, _p(means2d), _p(conics), _p(feats), _p(opac), _p(offsets), _p(ids),
                         _p(render), _p(alphas), _p(last))  # fmt: skip
    return render, alphas, last


def raster_bwd(means2d, conics, feats, opac, width, height, ts, offsets, ids, alphas, last, v_render, v_alphas):
    N, C = feats.shape
    means2d, conics, feats, opac, alphas, v_render, v_alphas = map(
        _c, (means2d, conics, feats, opac, alphas, v_render, v_alphas)
    )
    out = [torch.empty(N, 2), torch.empty(N, 2), torch.empty(N, 3), torch.empty(N, C), torch.empty(N)]
    lib().fgo_raster_bwd(N, C, width, height, ts, _p(means2d), _p(conics), _p(feats), _p(opac), _p(offsets), _p(ids),
                         _p(alphas), _p(last), _p(v_render), _p(v_alphas), *[_p(o) for o in out])  # fmt: skip
    return out


class _CompositeC(torch.autograd.Function):
    """K5 / K6 of the C restatement as one autograd node, so that the torch oracle's projection and
    SH stages (autograd) can be chained with per-pixel sequential C compositing."""

    @staticmethod
    def forward(ctx, means2d, conics, feats, opac, geom, offsets, ids, holder):
        width, height, ts = geom
        render, alphas, last = raster_fwd(means2d, conics, feats, opac, width, height, ts, offsets, ids)
        ctx.save_for_backward(means2d, conics, feats, opac, offsets, ids, alphas, last)
        ctx.geom, ctx.holder = geom, holder
        ctx.mark_non_differentiable(last)
        return render, alphas, last

    @staticmethod
    def backward(ctx, v_render, v_alphas, _v_last):
        means2d, conics, feats, opac, offsets, ids, alphas, last = ctx.saved_tensors
        width, height, ts = ctx.geom
        v_xy, v_abs, v_conic, v_col, v_op = raster_bwd(means2d, conics, feats, opac, width, height, ts, offsets, ids,
                                                       alphas, last, v_render, v_alphas)  # fmt: skip
        if ctx.holder is not None:
            ctx.holder.absgrad = v_abs[None]
        return v_xy, v_conic, v_col, v_op, None, None, None, None


def composite(means2d, conics, feats, opac, width, height, ts, offsets, ids, absgrad_holder=None):
    """``compositor=`` argument of raster_oracle.rasterization: the C compositing, differentiable."""
    return _CompositeC.apply(means2d, conics, feats, opac, (int(width), int(height), int(ts)),
                             offsets.to(torch.int32).contiguous(), ids.to(torch.int32).contiguous(), absgrad_holder)  # fmt: skip
