"""freegaussian_amd -- MI355X-native Gaussian raster path behind freegaussian's plugin surface.

The product is the HIP library ``libfgraster.so`` (C ABI in ``include/fgraster.h``); this
package is the thin Python host that mirrors the reference's operator interface:

    from freegaussian_amd import rasterization, quat_to_rotmat, num_sh_bases

and, for the iteration around that call, the reference's image loss and the Gaussian groups' optimizer step as single
launches (``l1_ssim``, ``FusedAdam``).
"""
from .rasterization import num_sh_bases, quat_to_rotmat, rasterization, rasterize_gauss_params  # noqa: F401



def __getattr__(name):  # (lazy: ops / optim import torch.autograd machinery only when asked for)
    if name == "l1_ssim":
        from .ops import l1_ssim

        return l1_ssim
    if name == "FusedAdam":
        from .optim import FusedAdam

        return FusedAdam
    raise AttributeError(name)


__all__ = ["rasterization", "rasterize_gauss_params", "quat_to_rotmat", "num_sh_bases", "l1_ssim", "FusedAdam"]
