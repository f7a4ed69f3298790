"""What would two independent 32-pixel half-strips per wavefront buy the raster backward?  (VERDICT r2 item 2,
DESIGN section 6 item 1.)  On the bench scene, from the lists the compositing walked and the pixels' last
contributing entries, count per tile the loop iterations (one iteration = one evaluation of a 64-lane slot,
~44 vector instructions in the backward) of
  now    : one wavefront per tile, slots = 16x4 strips, an entry costs one iteration per strip with a live pixel;
  rounds : left / right 8-column halves walk their OWN filtered lists in lockstep, the k-th entry of the left
           list beside the k-th of the right one; a pair costs max(live 8x4 cells left, right) iterations and
           ONE combined reduction (the variant sketched in DESIGN section 6);
  flat   : the two halves run through their (entry, 8x4 cell) items independently, no pairing: max over the
           halves of the item counts (needs the pixel state in LDS and per-half reductions: the upper bound of
           what any such scheme can reach).
Prints iterations relative to `now` and live lanes per iteration."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

sc = synthetic_scene(1_000_000, 1920, 1080, n_views=8, seed=42)
dev = torch.device("cuda", 0)
t = [x.to(dev) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
with torch.no_grad():
    r, a, info = rasterization(*t, sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), 1920, 1080, sh_degree=3, packed=False)
offs, ids = info["raster_isect_offsets"].long(), info["raster_flatten_ids"].long()
last = info["last_ids"].long()
m2, con, op = info["means2d"][0], info["conics"][0], info["opacities"][0]
tw = info["tile_width"]
g = torch.Generator().manual_seed(0)
tiles = torch.randint(0, offs.numel() - 1, (600,), generator=g)
yy, xx = torch.meshgrid(torch.arange(16, device=dev), torch.arange(16, device=dev), indexing="ij")
tot = dict(now=0, rounds=0, flat=0, lanes=0, reductions_now=0, reductions_rounds=0, reductions_flat=0)
for tile in tiles.tolist():
    s, e = int(offs[tile]), int(offs[tile + 1])
    if e <= s:
        continue
    ty, tx = divmod(tile, tw)
    gi = ids[s:e]
    px = tx * 16 + xx.reshape(-1).float() + 0.5
    py = ty * 16 + yy.reshape(-1).float() + 0.5
    dx, dy = m2[gi, 0:1] - px[None], m2[gi, 1:2] - py[None]
    sig = 0.5 * (con[gi, 0:1] * dx * dx + con[gi, 2:3] * dy * dy) + con[gi, 1:2] * dx * dy
    alpha = torch.clamp(op[gi, None] * torch.exp(-sig), max=0.999)
    lid = last[ty * 16 : ty * 16 + 16, tx * 16 : tx * 16 + 16]
    if lid.shape != (16, 16):
        continue  # (edge tiles: skipped in this estimate)
    reached = torch.arange(s, e, device=dev)[:, None] <= lid.reshape(1, -1)
    valid = ((sig >= 0) & (alpha >= 1.0 / 255.0) & reached).reshape(-1, 16, 16)  # [entries, y, x]
    strips = valid.reshape(-1, 4, 64).any(-1)  # [entries, 4]
    cells_now = strips.sum(-1)
    tot["now"] += int(cells_now.sum())
    tot["lanes"] += int(valid.sum())
    tot["reductions_now"] += int((cells_now > 0).sum())
    halves = valid.reshape(-1, 4, 4, 2, 8).any(2).any(-1)  # [entries, strip, half]: live 8x4 cells
    ca, cb = halves[:, :, 0].sum(-1), halves[:, :, 1].sum(-1)
    la, lb = ca[ca > 0], cb[cb > 0]
    n = min(la.numel(), lb.numel())
    tot["rounds"] += int(torch.maximum(la[:n], lb[:n]).sum()) + int(la[n:].sum()) + int(lb[n:].sum())
    tot["reductions_rounds"] += max(la.numel(), lb.numel())
    tot["flat"] += max(int(la.sum()), int(lb.sum()))
    tot["reductions_flat"] += la.numel() + lb.numel()
print(tot)
for k in ("now", "rounds", "flat"):
    print(f"{k:7s}: {tot[k] / tot['now']:.3f} x iterations, {tot['lanes'] / tot[k]:.1f} live lanes per iteration, "
          f"{tot['reductions_' + k] / tot['reductions_now']:.2f} x reductions")
