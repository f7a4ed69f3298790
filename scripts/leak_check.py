import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from freegaussian_amd import rasterization
from freegaussian_amd.scenes import synthetic_scene
from freegaussian_amd.viewdp import FlatGaussianParams
sc = synthetic_scene(200000, 960, 540, n_views=8, seed=42)
dev = torch.device("cuda", 0)
p = FlatGaussianParams.from_scene(sc, dev)
vr = torch.randn(1, 540, 960, 3, device=dev)
def step(v):
    vm, K = sc.viewmats[v:v+1].to(dev), sc.Ks[v:v+1].to(dev)
    with p.direct_grads():
        r, a, info = rasterization(*p.raster_inputs(), vm, K, 960, 540, sh_degree=3, packed=False, absgrad=True)
        r.backward(vr)
for i in range(20): step(i % 8)
torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); r0 = torch.cuda.memory_reserved()
for i in range(600): step(i % 8)
torch.cuda.synchronize(); m1 = torch.cuda.memory_allocated(); r1 = torch.cuda.memory_reserved()
print(f"allocated {m0/1e6:.1f} -> {m1/1e6:.1f} MB, reserved {r0/1e6:.1f} -> {r1/1e6:.1f} MB after 600 steps over 8 views")
