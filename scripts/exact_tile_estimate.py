"""How many (Gaussian, tile) pairs of the footprint rectangles does an EXACT ellipse-vs-tile test drop?
Bench scene family (CPU, oracle projection), subsample of the Gaussians at full resolution.
Pair kept iff min over the tile's pixel-centre box of d^T Q d <= 2 ln(255 o)  (some pixel centre region reaches
alpha >= 1/255; box relaxation of the pixel grid: conservative).
Usage: python scripts/exact_tile_estimate.py [n] [W] [H] [layout] [view]   (layout: freegaussian_amd.scenes.apply_layout)"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd.scenes import apply_layout, synthetic_scene  # noqa: E402
from oracle import raster_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
layout = sys.argv[4] if len(sys.argv) > 4 else "uniform"
view = int(sys.argv[5]) if len(sys.argv) > 5 else 4
sc = apply_layout(synthetic_scene(1_000_000, W, H, n_views=8, sh_degree=3, seed=42), layout)
idx = torch.randperm(1_000_000, generator=torch.Generator().manual_seed(0))[:n]
p = O.project(sc.means[idx], sc.quats[idx], sc.scales[idx], sc.viewmats[view], sc.Ks[view], W, H)
vis = p.radii > 0
mu, Q, o, rad = p.means2d[vis], p.conics[vis], sc.opacities[idx][vis], p.radii[vis].float()
tau = 2 * torch.log(255 * o)
keep = tau > 0
mu, Q, o, rad, tau = mu[keep], Q[keep], o[keep], rad[keep], tau[keep]
a, b, c = Q[:, 0], Q[:, 1], Q[:, 2]
det = a * c - b * b
ex, ey = torch.sqrt(tau * c / det), torch.sqrt(tau * a / det)  # AABB half-widths of the alpha >= 1/255 ellipse
ex, ey = torch.minimum(ex, rad), torch.minimum(ey, rad)
tw, th = (W + 15) // 16, (H + 15) // 16
# footprint rectangle: tiles whose pixel-centre span [16 t + 0.5, 16 t + 15.5] meets [mu - e, mu + e]
x0 = torch.clamp(torch.ceil((mu[:, 0] - ex - 15.5) / 16), 0, tw).long()
x1 = torch.clamp(torch.floor((mu[:, 0] + ex - 0.5) / 16) + 1, 0, tw).long()
y0 = torch.clamp(torch.ceil((mu[:, 1] - ey - 15.5) / 16), 0, th).long()
y1 = torch.clamp(torch.floor((mu[:, 1] + ey - 0.5) / 16) + 1, 0, th).long()
w, h = (x1 - x0).clamp_min(0), (y1 - y0).clamp_min(0)
area = w * h
tot = int(area.sum())
gid = torch.repeat_interleave(torch.arange(area.numel()), area)
k = torch.arange(tot) - torch.repeat_interleave(torch.cumsum(area, 0) - area, area)
tx = x0[gid] + k % w[gid].clamp_min(1)
ty = y0[gid] + k // w[gid].clamp_min(1)
# min of the quadratic form over the box [bx0, bx1] x [by0, by1] (pixel centres)
bx0, bx1 = 16.0 * tx + 0.5 - mu[gid, 0], 16.0 * tx + 15.5 - mu[gid, 0]
by0, by1 = 16.0 * ty + 0.5 - mu[gid, 1], 16.0 * ty + 15.5 - mu[gid, 1]
A, B, C, T = a[gid], b[gid], c[gid], tau[gid]


def qf(dx, dy):
    return A * dx * dx + 2 * B * dx * dy + C * dy * dy


inside = (bx0 <= 0) & (bx1 >= 0) & (by0 <= 0) & (by1 >= 0)
best = torch.full_like(A, float("inf"))
for xe in (bx0, bx1):  # vertical edges: x fixed, minimise over y
    y = torch.minimum(torch.maximum(-B * xe / C, by0), by1)
    best = torch.minimum(best, qf(xe, y))
for ye in (by0, by1):
    x = torch.minimum(torch.maximum(-B * ye / A, bx0), bx1)
    best = torch.minimum(best, qf(x, ye))
hit = inside | (best <= T)
ref_pairs = int((((torch.clamp(torch.floor((mu[:, 0] + rad) / 16) + 1, 0, tw) - torch.clamp(torch.floor((mu[:, 0] - rad) / 16), 0, tw)).clamp_min(0))
                 * ((torch.clamp(torch.floor((mu[:, 1] + rad) / 16) + 1, 0, th) - torch.clamp(torch.floor((mu[:, 1] - rad) / 16), 0, th)).clamp_min(0))).sum())
print(f"layout {layout} view {view}: radius-box pairs (the reference's lists, alpha-capable Gaussians) {ref_pairs}")
print(f"visible {int(vis.sum())} of {n}; footprint pairs {tot} ({tot / int(keep.sum()):.2f} per Gaussian); exact test keeps {int(hit.sum())} = {float(hit.float().mean()):.3f}")
sw = (tw + 1) // 2
st_rect = ((x1 - 1) // 2 - x0 // 2 + 1).clamp_min(0) * ((y1 - 1) // 2 - y0 // 2 + 1).clamp_min(0) * (area > 0)
st_id = (ty // 2) * sw + tx // 2
st_exact = torch.unique(gid[hit] * (sw * ((th + 1) // 2)) + st_id[hit]).numel()
print(f"supertile pairs: rectangles {int(st_rect.sum())}, exact {st_exact} = {st_exact / int(st_rect.sum()):.3f}")
