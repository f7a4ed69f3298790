"""cProfile of the host side of one fwd+bwd step at a small (host-bound) size."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from freegaussian_amd.viewdp import FlatGaussianParams  # noqa: E402

sc = synthetic_scene(100000, 480, 270, n_views=8, sh_degree=3, seed=42)
dev = torch.device("cuda", 0)
params = FlatGaussianParams.from_scene(sc, dev)
vm, K = sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev)
vr = torch.randn(1, 270, 480, 3, device=dev)


def step():
    with params.direct_grads():
        r, a, info = rasterization(*params.raster_inputs(), vm, K, 480, 270, sh_degree=3, render_mode="RGB",
                                   packed=False, absgrad=True)
        r.backward(vr)


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats(os.environ.get("FG_PROF_SORT", "tottime")).print_stats(int(os.environ.get("FG_PROF_N", "28")))
