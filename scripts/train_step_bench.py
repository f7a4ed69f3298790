"""harness.train_step on the bench scene: the whole training iteration a user of the reference runs (step_cb, get_outputs,
the reference's loss, backward, the six Adam optimizers, after_train_iter; no refinement inside the timed steps), and
where its time goes.  Usage: python scripts/train_step_bench.py [n_gauss] [steps] [width] [height]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import harness  # noqa: E402
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
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
opts = harness.build_optimizers(model)
c2w = torch.linalg.inv(sc.viewmats[0])
c2w[:3, 1:3] *= -1
K = sc.Ks[0]
cam = Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, times=torch.tensor([[0.0]]))
gt = torch.rand(H, W, 3, device=dev)
step0 = 3001  # SH degree 3; no refinement: num_train_data is not passed


graphed = None
if os.environ.get("FG_GRAPHED"):  # get_outputs + loss + backward as one hipGraph replay where the shape allows it
    from freegaussian_amd.graphed import GraphedModelStep

    graphed = GraphedModelStep(model, harness.main_loss)


def run(k, metrics_every=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        harness.train_step(model, opts, cam, gt, step0 + i, metrics_every=metrics_every, graphed=graphed)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


run(10)
total = run(steps)
total_log10 = run(steps, metrics_every=10)  # loss / psnr read back every 10th step only
# parts (each followed by a synchronize, so they add up to more than the pipelined step)
def timed(fn, k=steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def fwd_bwd():
    for o in opts.values():
        o.zero_grad(set_to_none=True)
    out = model.get_outputs(cam)
    ld = model.get_loss_dict(out, {"image": gt})
    (ld["main_loss"] + ld["scale_reg"]).backward()


def optim():
    for o in opts.values():
        o.step()


fwd_bwd()
res = {"graphed": bool(graphed is not None and graphed.applicable(cam)), "size": [n, W, H], "train_step_ms": round(total, 4), "train_step_ms_metrics_every_10": round(total_log10, 4), "outputs_loss_backward_ms": round(timed(fwd_bwd), 4),
       "optimizers_ms": round(timed(optim), 4), "after_train_iter_ms": round(timed(lambda: model.after_train_iter(model.step)), 4),
       "optimizers": {k: type(o).__name__ for k, o in opts.items()}}
print(json.dumps(res))
