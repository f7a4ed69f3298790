"""How full are 64-pixel cells of a tile for the (splat, tile) pairs the raster kernels walk?  Compares the kernels' 16x4
strips with 8x8 quadrants: live cells per pair and lanes per live cell -- a lane is live in an entry when alpha >= 1/255
AND the pixel has not finished before it (the forward's own last_ids), i.e. what the backward's slots hold.
Usage: python scripts/cell_shape_estimate.py [layout]   (freegaussian_amd.scenes.apply_layout)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import apply_layout, synthetic_scene  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "uniform"
sc = apply_layout(synthetic_scene(1_000_000, 1920, 1080, n_views=8, seed=42), layout)
dev = torch.device("cuda", 0)
t = [x.to(dev) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
with torch.no_grad():
    r, a, info = rasterization(*t, sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), 1920, 1080, sh_degree=3, packed=False)
offs, ids = info["raster_isect_offsets"].long(), info["raster_flatten_ids"].long()  # the lists the kernels walk
last_ids = info["last_ids"].long()
m2, con, op = info["means2d"][0], info["conics"][0], info["opacities"][0]
tw = info["tile_width"]
g = torch.Generator().manual_seed(0)
tiles = torch.randint(0, offs.numel() - 1, (400,), generator=g)
tot = {"pairs": 0, "strip_cells": 0, "strip_lanes": 0, "quad_cells": 0, "quad_lanes": 0}
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
for tile in tiles.tolist():
    s, e = int(offs[tile]), int(offs[tile + 1])
    if e <= s:
        continue
    gi = ids[s:e]
    px = (tile % tw) * 16 + xx.reshape(-1).float() + 0.5
    py = (tile // tw) * 16 + yy.reshape(-1).float() + 0.5
    dx = m2[gi, 0:1] - px[None]
    dy = m2[gi, 1:2] - py[None]
    sig = 0.5 * (con[gi, 0:1] * dx * dx + con[gi, 2:3] * dy * dy) + con[gi, 1:2] * dx * dy
    alpha = torch.clamp(op[gi, None] * torch.exp(-sig), max=0.999)
    ty0, tx0 = (tile // tw) * 16, (tile % tw) * 16
    last = torch.full((16, 16), -1, dtype=torch.long, device=dev)
    lt = last_ids[ty0 : ty0 + 16, tx0 : tx0 + 16]
    last[: lt.shape[0], : lt.shape[1]] = lt
    reached = torch.arange(s, e, device=dev)[:, None, None] <= last[None]  # the pixel had not finished before the entry
    valid = ((sig >= 0) & (alpha >= 1.0 / 255.0)).reshape(-1, 16, 16) & reached  # [entries, y, x]
    strips = valid.reshape(-1, 4, 4 * 16).sum(-1)  # 4 strips of 4 rows
    quads = valid.reshape(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).sum(-1)
    tot["pairs"] += valid.shape[0]
    tot["strip_cells"] += int((strips > 0).sum()); tot["strip_lanes"] += int(strips.sum())
    tot["quad_cells"] += int((quads > 0).sum()); tot["quad_lanes"] += int(quads.sum())
print(layout, tot)
print(f"16x4 strips : {tot['strip_cells'] / tot['pairs']:.2f} live cells per pair, {tot['strip_lanes'] / tot['strip_cells']:.1f} lanes per live cell")
print(f"8x8 quadrants: {tot['quad_cells'] / tot['pairs']:.2f} live cells per pair, {tot['quad_lanes'] / tot['quad_cells']:.1f} lanes per live cell")
