"""Heavy-tile forward against the serial walk on the two-cluster test scene: where do they differ?"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

DEV = torch.device("cuda", 0)
sc = synthetic_scene(60_000, 1920, 1080, n_views=2, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
sc.means[:20_000] = sc.means[:20_000] * 0.1 + torch.tensor([-0.9, 0.3, 0.0])
sc.means[20_000:40_000] = sc.means[20_000:40_000] * 0.1 + torch.tensor([0.8, -0.2, 0.0])
sc.opacities[20_000:40_000] *= 0.04
ctx = ops.default_context
ctx.heavy_tile_len = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
outs = {}
for mode in ("never", "always"):
    ctx.heavy_tiles = mode
    t = [x.to(DEV).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    r, a, info = rasterization(*t, sc.viewmats[1:2].to(DEV), sc.Ks[1:2].to(DEV), 1920, 1080, sh_degree=3, packed=False, absgrad=True)
    (r * torch.ones_like(r)).sum().backward()
    torch.cuda.synchronize()
    outs[mode] = (r.detach()[0], a.detach()[0], info["last_ids"], [x.grad for x in t], info)
print("heavy calls", ctx.heavy_calls)
(r0, a0, l0, g0, info), (r1, a1, l1, g1, _) = outs["never"], outs["always"]
offs = info["raster_isect_offsets"].reshape(-1).cpu()
lens = torch.diff(offs)
tw = info["tile_width"]
d = (r1 - r0).abs().amax(-1)  # [H, W]
H, W = d.shape
dt = torch.nn.functional.max_pool2d(d[None, None], 16, ceil_mode=True)[0, 0].cpu()  # per tile
bad = (dt > 1e-5).nonzero()
print("max diff", float(d.max()), "tiles differing", bad.shape[0], "heavy tiles", int((lens > ctx.heavy_tile_len).sum()))
for ty, tx in bad[:20].tolist():
    tile = ty * tw + tx
    blk = d[ty * 16 : ty * 16 + 16, tx * 16 : tx * 16 + 16]
    rows = (blk > 1e-5).any(1).nonzero().flatten().tolist()
    print(f" tile {tile} ({tx},{ty}) len {int(lens[tile])} maxdiff {float(dt[ty, tx]):.3e} bad px {int((blk > 1e-5).sum())} rows {rows[:16]}",
          "alpha diff", float((a1 - a0).abs()[ty * 16 : ty * 16 + 16, tx * 16 : tx * 16 + 16].max()),
          "last diff", int((l1 != l0)[ty * 16 : ty * 16 + 16, tx * 16 : tx * 16 + 16].sum()))
print("last_ids mismatches", int((l1 != l0).sum()))
for n, x, y in zip(("means", "quats", "scales", "opac", "colors"), g1, g0):
    print(n, float((x - y).norm() / y.norm()))
