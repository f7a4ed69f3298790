"""Debug aid: what the raster forward reports in walk_out against what the strips really walked (last_ids)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from freegaussian_amd import ops  # noqa: E402
from freegaussian_amd.rasterization import rasterize_gauss_params  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

DEV = "cuda"
W, H = 1920, 1080
sc = synthetic_scene(60_000, W, H, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
sc.means[:30_000] = sc.means[:30_000] * 0.1 + torch.tensor([0.8, -0.2, 0.0])
sc.opacities[:30_000] *= 0.04
ctx = ops.RasterContext(env={})
raw = dict(means=sc.means, quats=sc.quats, log_scales=sc.scales.log(), opacity_logits=torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4)),
           features_dc=sc.colors[:, 0, :].contiguous(), features_rest=sc.colors[:, 1:, :].contiguous())
t = {k: v.to(DEV).requires_grad_(True) for k, v in raw.items()}
vm, K = sc.viewmats[1:2].to(DEV), sc.Ks[1:2].to(DEV)
with ops.use(ctx):
    for i in range(5):
        nxt = ops._count_ring_next
        r, a, info = rasterize_gauss_params(t["means"], t["quats"], t["log_scales"], t["opacity_logits"], t["features_dc"],
                                            t["features_rest"], vm, K, W, H, 3)
        torch.cuda.synchronize()
        offs = info["raster_isect_offsets"].reshape(-1)
        last = info["last_ids"].reshape(H, W)
        tiles = last.new_zeros(H // 16 + 1, W // 16)
        walked = 0
        ly = (torch.arange(H, device=DEV) // 16)[:, None].expand(H, W)
        lx = (torch.arange(W, device=DEV) // 16)[None, :].expand(H, W)
        start = offs[:-1].reshape(-1)[(ly * (W // 16) + lx).reshape(-1)].reshape(H, W)
        walked = int((last - start + 1).max())
        words = ops._count_ring_np[ops._RING_WORDS * nxt : ops._RING_WORDS * nxt + 16]
        print(f"call {i}: slot {nxt} words {list(words)}; heavy_calls {ctx.heavy_calls}; longest list {int(torch.diff(offs).max())}; "
              f"longest walk of a pixel {walked}; long_walks {dict(ctx.long_walks)}; heavy_shapes {dict(ctx.heavy_shapes)}")
        del r, a
