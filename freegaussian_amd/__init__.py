"""freegaussian_amd -- MI355X-native Gaussian raster path behind freegaussian's plugin surface.

The product is the HIP library ``libfgraster.so`` (C ABI in ``include/fgraster.h``); this
package is the thin Python host that mirrors the reference's operator interface:

    from freegaussian_amd import rasterization, quat_to_rotmat, num_sh_bases
"""
from .rasterization import num_sh_bases, quat_to_rotmat, rasterization, rasterize_gauss_params  # noqa: F401

__all__ = ["rasterization", "rasterize_gauss_params", "quat_to_rotmat", "num_sh_bases"]
