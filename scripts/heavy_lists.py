"""The heavy tiles' job lists on a clustered scene: how many tiles are heavy, how many local / combine jobs per XCD."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import _lib, ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

frac, ball = float(sys.argv[1]), float(sys.argv[2])
sc = synthetic_scene(1_000_000, 1920, 1080, n_views=1, sh_degree=3, seed=42)
sc.means[: int(frac * 1_000_000)] *= ball / 2.0
dev = torch.device("cuda", 0)
ctx = ops.default_context
ctx.heavy_tiles = "always"
t = [x.to(dev).requires_grad_(True) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
for _ in range(2):
    r, a, info = rasterization(*t, sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), 1920, 1080, sh_degree=3, packed=False, absgrad=True)
offs = info["raster_isect_offsets"]
jobs = offs._fg_jobs[0][0].cpu()
lens = torch.diff(offs.reshape(-1).cpu())
cfgp = offs._fg_jobs[3]
words = int(_lib.load().fg_raster_jobs_words(1920, 1080, 16, cfgp))
LOCAL_WORDS, HEAVY_WORDS = 8 + 8 * 8192, 8 + 8 * 2048
main_words = words - LOCAL_WORDS - HEAVY_WORDS - 120 * 68  # (behind them: the table of first checkpoint slots, a word per tile)
cap = (main_words - 8) // 8
print("main jobs per XCD", jobs[:8].tolist())
print("local jobs per XCD", jobs[main_words : main_words + 8].tolist())
print("combine jobs per XCD", jobs[main_words + LOCAL_WORDS : main_words + LOCAL_WORDS + 8].tolist())
heavy = set()
for x in range(8):
    n = int(jobs[main_words + LOCAL_WORDS + x])
    e = jobs[main_words + LOCAL_WORDS + 8 + x * 2048 : main_words + LOCAL_WORDS + 8 + x * 2048 + n]
    heavy |= set((e >> 3).tolist())
hl = sorted(int(lens[t_]) for t_ in heavy)
print("heavy tiles", len(heavy), "shortest", hl[:5], "longest", hl[-5:], "| tiles > 1536:", int((lens > 1536).sum()), "> 4096:", int((lens > 4096).sum()),
      "> 10000:", int((lens > 10000).sum()))
