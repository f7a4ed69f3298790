"""The supertile binning's two C-ABI calls alone on the bench scene's rectangles (no raster behind them), timed
with events; for A/B of library builds whose lists may be WRONG on purpose (FG_SB_DEBUG_SCATTER: scatter without its
stores / with linear stores) -- never run a raster launch on those.
Usage: [FG_RASTER_LIB=...] python scripts/stbin_micro.py [n_gauss] [width] [height] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import _lib, ops  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dev = torch.device("cuda", 0)
sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=42)
t = [x.to(dev) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
out = ops.preprocess(*t, None, sc.viewmats[0].to(dev), sc.Ks[0].to(dev), W, H, sh_degree=3)
keys, rects = out[-1]._fg_bin[:2]
lib = _lib.load()
tw, th = (W + 15) // 16, (H + 15) // 16
T = tw * th
ptr = ops._ptr
offs = torch.empty(T + 1, dtype=torch.int32, device=dev)
loffs = torch.empty(T + 1, dtype=torch.int32, device=dev)
ws1 = torch.empty(int(lib.fg_stbin_count_workspace_bytes(n, tw, th)), dtype=torch.uint8, device=dev)
s = ops._stream()
ops._call("fg_stbin_count", n, ptr(rects), tw, th, ptr(offs), None, ptr(ws1), ws1.numel(), s)
torch.cuda.synchronize()
total = int(offs[-1])
cap = int(total * 1.25)
ids = torch.empty(cap, dtype=torch.int32, device=dev)
ws2 = torch.empty(int(lib.fg_stbin_fill_workspace_bytes(cap)), dtype=torch.uint8, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tc = tf = 0.0
for r in range(reps + 5):
    ev[0].record()
    ops._call("fg_stbin_count", n, ptr(rects), tw, th, ptr(offs), None, ptr(ws1), ws1.numel(), s)
    ev[1].record()
    ops._call("fg_stbin_fill", n, ptr(keys), ptr(rects), tw, th, cap, ptr(offs), ptr(ws1), ptr(ids), ptr(loffs), ptr(ws2),
              ws2.numel(), s)
    ev[2].record()
    torch.cuda.synchronize()
    if r >= 5:
        tc += ev[0].elapsed_time(ev[1])
        tf += ev[1].elapsed_time(ev[2])
print(f"{os.environ.get('FG_RASTER_LIB', 'default').split('/')[-1]:28s} N={n} {W}x{H} I'={total} count {tc / reps * 1e3:7.1f} us  fill {tf / reps * 1e3:7.1f} us")
