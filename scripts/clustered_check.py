"""Raster timings on a NON-uniform scene: a fraction of the Gaussians is pulled into a small ball at
the centre, so a few hundred tiles hold lists many times the mean.  Usage:
python scripts/clustered_check.py [frac_in_ball] [ball_extent]   (env FG_RASTER_TAIL_* as usual)"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from freegaussian_amd.viewdp import FlatGaussianParams  # noqa: E402

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
ball = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
sc = synthetic_scene(1_000_000, 1920, 1080, n_views=1, sh_degree=3, seed=42)
n_in = int(frac * sc.means.shape[0])
sc.means[:n_in] *= ball / 2.0
dev = torch.device("cuda", 0)
params = FlatGaussianParams.from_scene(sc, dev)
vm, K = sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev)
vr = torch.randn(1, 1080, 1920, 3, device=dev)


def step():
    with params.direct_grads():
        r, a, info = rasterization(*params.raster_inputs(), vm, K, 1920, 1080, sh_degree=3, render_mode="RGB",
                                   packed=False, absgrad=True)
        r.backward(vr)
    return info


for _ in range(5):
    info = step()
torch.cuda.synchronize()
ops.default_context.stage_timer = ops.StageTimer()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20 * 1e3
st = ops.default_context.stage_timer.summary()
offs = info["raster_isect_offsets"].reshape(-1).long()  # [T + 1]: the lists the raster launches walked
lens = (offs[1:] - offs[:-1]).float()
print(json.dumps({"scene": [frac, ball], "I_raster": int(offs[-1]), "mean_len": float(lens.mean()), "max_len": float(lens.max()),
                  "step_ms": round(dt, 4), "fwd": st.get("fg_raster_fwd"), "bwd": st.get("fg_raster_bwd"),
                  "binning": os.environ.get("FG_BINNING", "supertile"), "long_segments": os.environ.get("FG_LONG_SEGMENTS", "auto"),
                  "long_calls": ops.default_context.long_calls, "stages": {k: round(v, 4) for k, v in st.items()}}))
