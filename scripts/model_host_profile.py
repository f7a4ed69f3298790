"""cProfile of the host side of FreeGaussianModel.get_outputs + backward at 1M / 1080p (the step is
host-bound when this takes longer than the ~1.0 ms of kernels).  Usage: python scripts/model_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = torch.device("cuda", 0)
sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=42)
cfg = FreeGaussianModelConfig(background_color="random", num_downscales=0, warm_up=10**9)
model = FreeGaussianModel(cfg, seed_points=sc.means, init_scales=-4.0)
with torch.no_grad():
    gp = model.gauss_params
    gp["scales"].copy_(sc.scales.log())
    gp["quats"].copy_(sc.quats)
    gp["opacities"].copy_(torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6))[:, None])
    gp["features_dc"].copy_(sc.colors[:, 0])
    gp["features_rest"].copy_(sc.colors[:, 1:])
model = model.to(dev).train()
model.step = 3000
c2w = torch.linalg.inv(sc.viewmats[0])
c2w[:3, 1:3] *= -1
K = sc.Ks[0]
cam = Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H,
             times=torch.tensor([[0.0]]))
vr = torch.randn(H, W, 3, device=dev)
params = list(model.gauss_params.values())


def step():
    for p in params:
        p.grad = None
    out = model.get_outputs(cam)
    (out["rgb"] * vr).sum().backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
ms = torch.cuda.memory_stats()
print({k: ms[k] for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated")})
