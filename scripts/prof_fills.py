import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from freegaussian_amd import rasterization
from freegaussian_amd.scenes import synthetic_scene
from freegaussian_amd.viewdp import FlatGaussianParams
from torch.profiler import ProfilerActivity, profile
sc = synthetic_scene(1000000, 1920, 1080, n_views=8, seed=42)
dev = torch.device("cuda", 0)
p = FlatGaussianParams.from_scene(sc, dev)
vm, K = sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev)
vr = torch.randn(1, 1080, 1920, 3, device=dev)
def step():
    with p.direct_grads():
        r, a, info = rasterization(*p.raster_inputs(), vm, K, 1920, 1080, sh_degree=3, packed=False, absgrad=True)
        r.backward(vr)
for _ in range(5): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::full"):
        print(e.name, e.input_shapes, round(e.device_time_total, 1))
