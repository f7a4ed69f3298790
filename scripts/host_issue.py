"""Host time to ISSUE one raster step (forward + backward) with an idle queue in front of it: every step is preceded by
a device synchronise, so nothing the host does waits for the GPU.  Splits out the time inside the two C-ABI calls
(fg_step_fwd / fg_step_bwd: the library's own kernel launches) from the Python around them.
Usage: python scripts/host_issue.py [n_gauss] [width] [height] [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import _lib, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from freegaussian_amd.viewdp import FlatGaussianParams  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 480
H = int(sys.argv[3]) if len(sys.argv) > 3 else 270
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=42)
dev = torch.device("cuda", 0)
params = FlatGaussianParams.from_scene(sc, dev)
vm, K = sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev)
vr = torch.randn(1, H, W, 3, device=dev)
lib = _lib.load()
inside = {"fg_step_fwd": 0.0, "fg_step_bwd": 0.0}


class Timed:
    """the library handle with the two step calls timed (attribute lookups of everything else pass through)"""

    def __getattr__(self, name):
        f = getattr(lib, name)
        if name not in inside:
            return f

        def g(*a):
            t = time.perf_counter()
            rc = f(*a)
            inside[name] += time.perf_counter() - t
            return rc

        return g


def step():
    with params.direct_grads():
        r, a, info = rasterization(*params.raster_inputs(), vm, K, W, H, sh_degree=3, render_mode="RGB", packed=False,
                                   absgrad=True)
        t1 = time.perf_counter()
        r.backward(vr)
    return t1


for _ in range(30):
    step()
torch.cuda.synchronize()
timed = Timed()
_lib.load = lambda: timed
for k in inside:
    inside[k] = 0.0
fwd = bwd = 0.0
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t1 = step()
    t2 = time.perf_counter()
    fwd += t1 - t0
    bwd += t2 - t1
torch.cuda.synchronize()
# the same loop without the synchronise: the host runs ahead of the queue as far as the step lets it (it waits for the
# list length of its own forward); on a scene small enough for the GPU to keep up this is the host's time per step
t0 = time.perf_counter()
for _ in range(steps):
    step()
pipelined = (time.perf_counter() - t0) / steps * 1e3
torch.cuda.synchronize()
ms = lambda x: round(x / steps * 1e3, 4)
print(json.dumps({"size": [n, W, H], "step_calls": bool(inside["fg_step_fwd"]), "host_issue_ms": ms(fwd + bwd), "pipelined_ms_per_step": round(pipelined, 4),
                  "forward_ms": ms(fwd), "backward_ms": ms(bwd),
                  "inside_fg_step_fwd_ms": ms(inside["fg_step_fwd"]), "inside_fg_step_bwd_ms": ms(inside["fg_step_bwd"])}))
