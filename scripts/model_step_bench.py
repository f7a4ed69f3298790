"""Model-level step (FreeGaussianModel.get_outputs + backward, no deform MLP) on the bench scene:
what a user of the reference model sees around the raster call (§8f row 3).
Usage: python scripts/model_step_bench.py [n_gauss] [steps] [width] [height]
With width x height at or below graphed.GraphedModelStep's tile limit the same step is also timed as one
hipGraph replay (`graphed_model_step_ms`): the reference's first 6000 steps run at 1/4 and 1/2 resolution."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops  # noqa: E402
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
dev = torch.device("cuda", 0)
sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=42)
cfg = FreeGaussianModelConfig(background_color="random", num_downscales=0, warm_up=int(os.environ.get("FG_MODEL_WARM_UP", 10**9)),  # (0: the deformation MLP runs)
                              fused_front_end=not os.environ.get("FG_UNFUSED"))
model = FreeGaussianModel(cfg, seed_points=sc.means, init_scales=-4.0)
with torch.no_grad():
    gp = model.gauss_params
    gp["scales"].copy_(sc.scales.log())
    gp["quats"].copy_(sc.quats)
    gp["opacities"].copy_(torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6))[:, None])
    gp["features_dc"].copy_(sc.colors[:, 0])
    gp["features_rest"].copy_(sc.colors[:, 1:])
model = model.to(dev).train()
model.step = int(os.environ.get("FG_MODEL_STEP", "3000"))  # SH degree = step // 1000, 3 from step 3000 on
w2c = sc.viewmats[0]
c2w = torch.linalg.inv(w2c)
c2w[:3, 1:3] *= -1  # OpenCV -> OpenGL (get_viewmat flips back)
K = sc.Ks[0]
cam = Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H,
             times=torch.tensor([[0.0]]))
vr = torch.randn(H, W, 3, device=dev)
params = list(model.gauss_params.values())


# FG_MODEL_LOSS=main: the reference's loss (get_loss_dict: L1 + SSIM against a ground-truth image) instead of a plain
# weighted sum; =torch: the same loss through torch operators (harness.ssim), as it ran before csrc/loss.hip
loss_kind = os.environ.get("FG_MODEL_LOSS", "")
gt_img = torch.rand(H, W, 3, device=dev)
if loss_kind == "torch":
    from freegaussian_amd import harness

    harness.l1_and_ssim = lambda pred, gt: ((gt - pred).abs().mean(),
                                            harness.ssim(gt.permute(2, 0, 1)[None], pred.permute(2, 0, 1)[None]))


def step():
    for p in params:
        p.grad = None
    out = model.get_outputs(cam)
    if loss_kind:
        ld = model.get_loss_dict(out, {"image": gt_img})
        (ld["main_loss"] + ld["scale_reg"]).backward()
    else:
        (out["rgb"] * vr).sum().backward()


for _ in range(25):  # (the caching allocator and the list-capacity history settle over the first steps)
    step()
# host-bound sizes follow the host's clock and whatever else the box is doing: several blocks, median and best reported;
# FG_MODEL_AB=1 alternates blocks with and without the one-call-per-direction path (ops.RasterContext.step_calls)
blocks = int(os.environ.get("FG_MODEL_BLOCKS", "5"))
ab = bool(os.environ.get("FG_MODEL_AB"))
block_ms = {True: [], False: []}
issue_ms = []
for b in range(blocks * (2 if ab else 1)):
    calls = ops.default_context.step_calls if not ab else (b % 2 == 0)
    saved, ops.default_context.step_calls = ops.default_context.step_calls, calls
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    issue_ms.append((time.perf_counter() - t0) / steps * 1e3)  # host time to enqueue a step (includes the one host sync)
    torch.cuda.synchronize()
    block_ms[bool(calls)].append((time.perf_counter() - t0) / steps * 1e3)
    ops.default_context.step_calls = saved
cur = block_ms[bool(ops.default_context.step_calls)]
dt, issue = sorted(cur)[len(cur) // 2], sorted(issue_ms)[len(issue_ms) // 2]
# stage times from a separate pass: the HIP events of the timer (two per C-ABI call) cost more host time than the
# step itself at these sizes
ops.default_context.stage_timer = ops.StageTimer()
for _ in range(steps):
    step()
stages = ops.default_context.stage_timer.summary()
ops.default_context.stage_timer = None
graphed_ms = None
from freegaussian_amd.graphed import GraphedModelStep  # noqa: E402

gstep = GraphedModelStep(model, lambda rgb, gt: (rgb * gt).sum())
if gstep.applicable(cam):
    for p in params:
        p.grad = None
    for _ in range(5):
        gstep.step(cam, vr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gstep.step(cam, vr)
    torch.cuda.synchronize()
    graphed_ms = (time.perf_counter() - t0) / steps * 1e3
res = {"loss": loss_kind or "weighted sum", "model_step_ms": dt, "model_step_ms_best": min(cur),
       "blocks_ms": {("step_calls" if k else "stage_wise"): [round(x, 4) for x in v] for k, v in block_ms.items() if v}, "graphed_model_step_ms": graphed_ms, "size": [n, W, H], "host_issue_ms": issue, "fused_front_end": cfg.fused_front_end, "hip_stage_ms": sum(stages.values()), "stages": {k: round(v, 4) for k, v in stages.items()}}
if os.environ.get("FG_MODEL_PROFILE"):
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:25]
    res["top_device_ops_us_per_step"] = {e.key[:70]: round(e.device_time_total / 5, 1) for e in rows if e.device_time_total > 0}
print(json.dumps(res, indent=1))
