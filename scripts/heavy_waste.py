"""Heavy tiles: how much of the local pass's work lies behind the point where the tile's last pixel stops?
Per view of the bench's clustered layout: entries the local jobs composite (every entry behind the 2048-entry prefix of
every heavy tile that is still open there) against entries up to the tile's largest last_ids.
Usage: python scripts/heavy_waste.py [frac] [extent]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
ball = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
sc = synthetic_scene(1_000_000, 1920, 1080, n_views=8, sh_degree=3, seed=42)
sc.means[: int(frac * 1_000_000)] *= ball / 2.0
dev = torch.device("cuda", 0)
ins = [t.to(dev) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
PREFIX, THR = 2048, ops.default_context.heavy_tile_len
for v in range(8):
    with torch.no_grad():
        for _ in range(2):
            r, a, info = rasterization(*ins, sc.viewmats[v:v + 1].to(dev), sc.Ks[v:v + 1].to(dev), 1920, 1080, sh_degree=3)
    offs = info["raster_isect_offsets"].reshape(-1).long()
    start, lens = offs[:-1], offs[1:] - offs[:-1]
    last = info["last_ids"].long()  # [H, W]
    H, W = last.shape
    th, tw = (H + 15) // 16, (W + 15) // 16
    pad = torch.full((th * 16, tw * 16), -1, device=dev, dtype=torch.long)
    pad[:H, :W] = last
    tile_last = pad.view(th, 16, tw, 16).permute(0, 2, 1, 3).reshape(th * tw, 256).max(dim=1).values  # largest index used
    used = (tile_last - start + 1).clamp(min=0)
    heavy = lens > THR
    open_ = heavy & (used > PREFIX)
    local = (lens[open_] - PREFIX).sum().item()
    needed = (used[open_] - PREFIX).sum().item()
    print(json.dumps({"view": v, "entries": int(offs[-1]), "heavy_tiles": int(heavy.sum()), "open_behind_prefix": int(open_.sum()),
                      "entries_local_pass": int(local), "entries_up_to_last_stop": int(needed),
                      "longest_open": int(lens[open_].max().item()) if open_.any() else 0}))
