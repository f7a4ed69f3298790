"""Which fp32 projection backward is closer to fp64: the HIP kernel (fg_project_bwd, the code the fused
fg_preprocess_bwd shares) or torch autograd of the oracle's projection?  Bench scene, view 3, random
upstream gradients on means2d / depths / conics.  Usage: python scripts/project_bwd_accuracy.py [n]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from oracle import raster_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.set_num_threads(16)
sc = synthetic_scene(n, 1920, 1080, n_views=8, sh_degree=3, seed=42)
v = 3
g = torch.Generator().manual_seed(0)
v2d, vz, vc = torch.randn(n, 2, generator=g), torch.randn(n, generator=g), torch.randn(n, 3, generator=g)


def oracle(dt):
    ins = [t.clone().to(dt).requires_grad_(True) for t in (sc.means, sc.quats, sc.scales)]
    p = O.project(*ins, sc.viewmats[v].to(dt), sc.Ks[v].to(dt), 1920, 1080)
    ((p.means2d * v2d.to(dt)).sum() + (p.depths * vz.to(dt)).sum() + (p.conics * vc.to(dt)).sum()).backward()
    return [t.grad.double() for t in ins], p.radii


g64, r64 = oracle(torch.float64)
g32, r32 = oracle(torch.float32)
dev = torch.device("cuda", 0)
ins = [t.to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.scales)]
radii, m2d, dep, con, _, _ = ops.project(*ins, sc.viewmats[v].to(dev), sc.Ks[v].to(dev), 1920, 1080)
((m2d * v2d.to(dev)).sum() + (dep * vz.to(dev)).sum() + (con * vc.to(dev)).sum()).backward()
gh = [t.grad.double().cpu() for t in ins]
same = (r64 == r32) & (r32 == radii.cpu())  # rows culled identically by all three
print("rows compared", int(same.sum()), "of", n)
for name, a64, a32, ah in zip(("means", "quats", "scales"), g64, g32, gh):
    m = same[:, None].expand_as(a64)
    ref = a64[m]
    print(f"{name:7s} torch fp32 vs fp64: {float((a32[m] - ref).norm() / ref.norm()):.2e}   HIP vs fp64: {float((ah[m] - ref).norm() / ref.norm()):.2e}"
          f"   HIP vs torch fp32: {float((ah[m] - a32[m]).norm() / a32[m].norm()):.2e}")
