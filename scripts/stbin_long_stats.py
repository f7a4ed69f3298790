"""What the long-segment sample sort (csrc/stbin.hip, FG_STBIN_LONG_SEGMENTS) did on a clustered scene: long segments,
buckets, bucket-size distribution (read back from the two workspaces, whose layout this script mirrors).
Usage: python scripts/stbin_long_stats.py [frac] [extent] [n_gauss]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import _lib, ops  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
ball = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
W, H = 1920, 1080
sc = synthetic_scene(N, W, H, n_views=1, sh_degree=3, seed=42)
sc.means[: int(frac * N)] *= ball / 2.0
dev = torch.device("cuda", 0)
t = [x.to(dev) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
_, _, _, _, _, splats = ops.preprocess(*t, None, sc.viewmats[0].to(dev), sc.Ks[0].to(dev), W, H, sh_degree=3)
keys, rects = splats._fg_bin[:2]
lib = _lib.load()
tw, th = (W + 15) // 16, (H + 15) // 16
T, S = tw * th, ((tw + 1) // 2) * ((th + 1) // 2)
ws1 = torch.zeros(int(lib.fg_stbin_count_workspace_bytes(N, tw, th)), dtype=torch.uint8, device=dev)
offs = torch.zeros(T + 1, dtype=torch.int32, device=dev)
ops._call("fg_stbin_count", N, ops._ptr(rects), tw, th, ops._ptr(offs), None, ops._ptr(ws1), ws1.numel(), ops._stream())
cap = int(offs[-1]) + 1000
ws2 = torch.zeros(int(lib.fg_stbin_fill_workspace_bytes(cap)), dtype=torch.uint8, device=dev)
ids = torch.zeros(cap, dtype=torch.int32, device=dev)
lo = torch.zeros(T + 1, dtype=torch.int32, device=dev)
for rep in range(3):
    k2 = keys.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops._call("fg_stbin_fill", N, ops._ptr(k2), ops._ptr(rects), tw, th, cap, ops._ptr(offs), ops._ptr(ws1), ops._ptr(ids),
              ops._ptr(lo), ops._ptr(ws2), ws2.numel(), 1, ops._stream())
    e1.record()
    torch.cuda.synchronize()
print("fill ms", e0.elapsed_time(e1), "list", int(offs[-1]))


def al(x):
    return (x + 255) & ~255


nc = (N + 4095) // 4096
o = al(nc * T * 4) + al(nc * S * 4)
st_off = ws1[o : o + (S + 1) * 4].view(torch.int32).cpu().numpy()
o += al((S + 1) * 4) + al((S + 1) * 4)
ll = ws1[o : o + (S + 2) * 16].view(torch.int32).cpu().numpy().reshape(-1, 4)
L, chunks, buckets = ll[0][:3]
seg_n = st_off[1:] - st_off[:-1]
print("supertile elements", int(st_off[-1]), "long segments", L, "chunks", chunks, "buckets", buckets,
      "elements in long segments", int(ll[1 : 1 + L, 3].sum()), "largest", int(seg_n.max()))
kb = cap // 1536 + cap // 7936 + 2
o2 = 2 * al(cap * 8) + al(kb * 8) + al(kb * 16)
cnt = ws2[o2 : o2 + kb * 4].view(torch.int32).cpu().numpy()[:buckets]
o3 = o2 + al(kb * 4) + al(kb * 4) + al(kb * 16)
over = ws2[o3 : o3 + 8].view(torch.int32).cpu().numpy()
print("skewed whole segments left by the small launch:", over[0], " buckets beyond the LDS sort:", over[1])
print("bucket sizes: mean %.0f p50 %d p99 %d max %d; > 3072: %d; empty: %d" % (cnt.mean(), np.median(cnt), np.percentile(cnt, 99),
                                                                                cnt.max(), (cnt > 3072).sum(), (cnt == 0).sum()))
for i in range(1, 1 + min(L, 12)):
    st, cb, bb, n = ll[i]
    k = ll[i + 1][2] - bb
    c = cnt[bb : bb + k]
    print(f"  segment st {st} n {n} buckets {k}: max {c.max()} min {c.min()} sum {c.sum()}")
