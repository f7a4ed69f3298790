"""ctypes binding of ``libfgraster.so`` (the C ABI declared in ``include/fgraster.h``).

There is NO fallback: if the HIP library is missing or a symbol is absent, importing the ops
raises.  PyTorch is only used by callers for device memory and streams; nothing here takes a
torch type -- pointers are plain integers."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfgraster.so")

P = c_void_p

# name -> (restype, argtypes); must list every symbol of include/fgraster.h
SIGNATURES = {
    "fg_abi_version": (c_int, []),
    "fg_error_string": (c_char_p, [c_int]),
    "fg_project_fwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, c_float, c_int,
                               P, P, P, P, P, P, P]),
    "fg_project_bwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_float, P, P, P, P, P, P, P, P, P, P, P]),
    "fg_sh_fwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P]),
    "fg_sh_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "fg_scan_workspace_bytes": (c_size_t, [c_int]),
    "fg_scan_tiles": (c_int, [c_int, P, P, P, c_size_t, P]),
    "fg_tile_bin": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "fg_sort_workspace_bytes": (c_size_t, [c_int64]),
    "fg_sort_pairs": (c_int, [c_int64, P, P, c_int, P, c_size_t, P]),
    "fg_tile_ranges": (c_int, [c_int64, P, c_int, P, P]),
    "fg_sort32_workspace_bytes": (c_size_t, [c_int64]),
    "fg_sort_pairs32": (c_int, [c_int64, P, P, c_int, P, c_size_t, P]),
    "fg_bin_prepare_workspace_bytes": (c_size_t, [c_int]),
    "fg_bin_prepare": (c_int, [c_int, P, P, P, P, P, P, c_size_t, P]),
    "fg_bin_prepare_rects": (c_int, [c_int, P, P, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "fg_bin_prepare_keys": (c_int, [c_int, P, P, P, P, P, P, P, c_size_t, P]),
    "fg_bin_emit_workspace_bytes": (c_size_t, [c_int64]),
    "fg_bin_emit_sort": (c_int, [c_int, c_int64, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    "fg_bin_emit_sort_capacity": (c_int, [c_int, c_int64, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, c_size_t,
                                          P]),
    "fg_stbin_supported": (c_int, [c_int, c_int, c_int]),
    "fg_stbin_count_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fg_stbin_count": (c_int, [c_int, P, P, c_int, c_int, P, P, P, c_size_t, P]),
    "fg_stbin_fill_workspace_bytes": (c_size_t, [c_int64]),
    "fg_stbin_fill": (c_int, [c_int, P, P, P, c_int, c_int, c_int64, P, P, P, P, P, c_size_t, c_int, P]),
    "fg_stbin_fill_jobs": (c_int, [c_int, P, P, P, c_int, c_int, c_int64, P, P, P, P, P, c_size_t, c_int, c_int, c_int, P, P,
                                   c_int, P, c_int, P, P]),
    "fg_isect_keys": (c_int, [c_int64, P, P, P, P, P]),
    "fg_densify_stats": (c_int, [c_int, P, P, c_float, P, P, P, P]),
    "fg_adam_step_multi": (c_int, [c_int, P, P]),  # (count, fg_adam_tensor[count], stream)
    "fg_adam_step": (c_int, [c_int64, P, P, P, P, c_double, c_double, c_double, c_double, c_int64, P]),
    "fg_l1_ssim_workspace_floats": (c_size_t, [c_int, c_int, c_int]),
    "fg_l1_ssim_fwd": (c_int, [c_int, c_int, c_int, P, P, P, P, c_size_t, P, P]),
    "fg_l1_ssim_bwd": (c_int, [c_int, c_int, c_int, P, P, P, P, P, P]),
    "fg_pack_splats": (c_int, [c_int, c_int, P, P, P, P, P, P]),
    # (the pointer before the stream of every raster entry point is the `const fg_raster_config*`)
    "fg_raster_config_init": (None, [P]),
    "fg_raster_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P]),
    "fg_raster_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P]),
    "fg_raster_composite_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "fg_raster_composite_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, c_int, P, P, P, P, P, P, P, P]),
    "fg_raster_jobs_words": (c_int64, [c_int, c_int, c_int, P]),
    "fg_raster_build_jobs": (c_int, [c_int, c_int, c_int, P, P, P, c_int, P, P]),
    "fg_raster_jobs_fwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, P, P, P, P, c_int64, P, P, P]),
    "fg_raster_seg_ckpt_floats": (c_int64, [c_int, c_int, c_int, c_int, c_int64, P]),
    "fg_raster_jobs_bwd": (c_int, [c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "fg_unpack_grads": (c_int, [c_int, c_int, P, P, P, P, P, P, P]),
    "fg_preprocess_fwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, P, c_int, c_int,
                                  c_float, c_float, c_float, c_float, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fg_viewmat_bwd_workspace_bytes": (c_size_t, [c_int]),
    "fg_viewmat_bwd": (c_int, [c_int, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                               c_float, c_int, P, P, P, c_int, P, P, P, P, P, c_size_t, P]),
    "fg_sh_pack_fwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, P, P, P, P, P, P]),
    "fg_preprocess_bwd": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                                  c_float, c_int, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P]),
    "fg_preprocess_raw_fwd": (c_int, [c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, c_int, P, P, c_int,
                                      c_int, c_float, c_float, c_float, c_float, c_int, c_int, P, P, P, P, P, P, P,
                                      P, P, P, P, P]),
    "fg_preprocess_raw_bwd": (c_int, [c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                                      c_float, c_int, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, P, P]),
    "fg_preprocess_bwd_factored": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                                           c_float, c_int, P, P, P, c_int, P, P, P, P, P, P, P, c_int, P, P, P]),
    "fg_sh_grad_accumulate": (c_int, [c_int, c_int, c_int, c_int, P, P, c_int64, c_int, c_float, P, P]),
    "fg_payload_compact": (c_int, [c_int, c_int, P, P, c_int64, P, P]),
    "fg_payload_expand": (c_int, [c_int, c_int, c_int, P, c_int64, c_int64, P, c_int64, P]),
    "fg_step_layout_query": (c_int, [P, P, P]),
    "fg_step_fwd": (c_int, [P, P, P, P, P, P, P]),
    "fg_step_bwd": (c_int, [P, P, P, P, P, P]),
    "fg_sh_grad_accumulate_split": (c_int, [c_int, c_int, c_int, c_int, P, P, c_int64, c_int, c_float, P, P, P]),
    "fg_preprocess_raw_bwd_factored": (c_int, [c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int,
                                               c_float, c_int, P, P, P, c_int, P, P, P, P, P, P, P, P, P, c_int, P, P, P]),
    "fg_densify_flags": (c_int, [c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P,
                                 P, P, P, P]),
    "fg_densify_map": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "fg_gather_rows": (c_int, [c_int64, c_int, P, P, c_int64, P, P]),
    "fg_split_children": (c_int, [c_int, c_int, P, P, P, P, P, P]),
    "fg_mask_backproject": (c_int, [c_int, P, P, P, P, c_int, c_int, P, P, c_int, c_int, P, P]),
    "fg_camera_flow": (c_int, [c_int, c_int, P, P, P, P, P, P]),
    "fg_reprojection_flow": (c_int, [c_int, c_int, P, P, P, P, c_float, P, P]),
    "fg_flow_fwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P]),
    "fg_flow_bwd": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, P, P, P]),
}  # fmt: skip

# test hooks, not declared in the public header
_EXTRA = {"fg_debug_wave_reduce16": (c_int, [P, P, P])}

ABI_VERSION = 9
STBIN_LONG_SEGMENTS = 1  # FG_STBIN_LONG_SEGMENTS
STBIN_TEST_SMALL_SLABS = 4  # FG_STBIN_TEST_SMALL_SLABS (tests: the sample sort's overflow path)
STEP_NO_FOOTPRINT_MASKS = 2  # FG_STEP_NO_FOOTPRINT_MASKS
SH_JAC_FLOATS = 10  # FG_SH_JAC_FLOATS
_lib = None


class FgRasterError(RuntimeError):
    pass


class RasterConfig(ctypes.Structure):
    """``fg_raster_config`` of include/fgraster.h: the launch policy of the raster kernels, handed to every
    raster entry point (the library itself reads no environment variable).  Field order = the header's."""

    _fields_ = [(n, ctypes.c_int32) for n in (
        "size", "ppt_fwd", "ppt_bwd", "tile_order", "bands_nx", "tail4_fwd", "tail2_fwd", "tail4_bwd", "tail2_bwd",
        "split4_fwd", "split2_fwd", "split4_bwd", "split2_bwd", "use_liveness", "seg_parts", "seg_tail", "seg_parts2",
        "seg_tail2", "debug_only_xcd", "debug_k_mod", "balance_bands", "heavy_tiles", "seg_slots", "prio_fwd", "prio_bwd",
        "heavy_wide", "seg_fine")]  # fmt: skip

    FIELDS = tuple(n for n, _ in _fields_)[1:]

    @classmethod
    def defaults(cls) -> "RasterConfig":
        cfg = cls()
        load().fg_raster_config_init(ctypes.byref(cfg))
        return cfg

    def ptr(self) -> int:
        return ctypes.addressof(self)


STEP_BUFFERS = ("radii", "means2d", "depths", "conics", "comp", "tiles", "splats", "depth_keys", "tile_rects", "tile_masks", "sh_jac",
                "tile_offsets", "list_offsets", "flatten_ids", "jobs", "live", "seg_ckpt", "v_splats", "render", "alphas",
                "last_ids", "clamp_mask", "count_ws", "fill_ws")  # the FG_STEP_* enum of include/fgraster.h, in order
STEP_BUFFER = {n: i for i, n in enumerate(STEP_BUFFERS)}


class StepDesc(ctypes.Structure):
    """``fg_step_desc``: field order = the header's."""

    _fields_ = [(n, ctypes.c_int32) for n in (
        "size", "N", "width", "height", "tile_size", "raw", "sh_degree", "k_stored", "n_color", "with_depth", "n_extra",
        "antialiased", "n_clamp", "want_backward", "list_shares", "flags")] + [
        (n, ctypes.c_float) for n in ("eps2d", "near_plane", "far_plane", "radius_clip")] + [("capacity", ctypes.c_int64)]  # fmt: skip


class StepIO(ctypes.Structure):
    """``fg_step_io``: device pointers as integers."""

    _fields_ = [(n, ctypes.c_void_p) for n in (
        "means", "quats", "d_quats", "scales", "d_scales", "opacities", "colors", "features_rest", "extra", "viewmat", "K",
        "background", "count_out", "v_render", "v_alphas", "v_depths", "v_conics", "v_means", "v_quats", "v_d_quats",
        "v_scales", "v_d_scales", "v_opacities", "v_colors", "v_features_rest", "v_extra", "v_rgb")] + [
        ("v_rgb_floats", ctypes.c_int32), ("ev_raster_begin", ctypes.c_void_p), ("ev_raster_end", ctypes.c_void_p),
        ("ckpt_need_out", ctypes.c_void_p)]  # fmt: skip


class StepLayout(ctypes.Structure):
    """``fg_step_layout``."""

    _fields_ = [("keep_bytes", ctypes.c_int64), ("tmp_bytes", ctypes.c_int64),
                ("offset", ctypes.c_int64 * len(STEP_BUFFERS)), ("nbytes", ctypes.c_int64 * len(STEP_BUFFERS)),
                ("jobs_words", ctypes.c_int64), ("seg_ckpt_floats", ctypes.c_int64), ("channels", ctypes.c_int32)]  # fmt: skip


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FgRasterError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C freegaussian_amd/csrc`.  There is no CPU fallback."
        )
    # FG_RASTER_LIB: load another build of the same library (the instrumented `make stats` one)
    lib = ctypes.CDLL(os.environ.get("FG_RASTER_LIB", LIB_PATH))
    for name, (res, args) in {**SIGNATURES, **_EXTRA}.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    if lib.fg_abi_version() != ABI_VERSION:
        raise FgRasterError(f"ABI mismatch: library {lib.fg_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().fg_error_string(code).decode()
        raise FgRasterError(f"{what} failed: {msg} ({code})")
