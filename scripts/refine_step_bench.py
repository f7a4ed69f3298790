"""A step SERIES across refinements: harness.train_step on the bench scene with `refinement_after` every 100 steps, as the
reference trains (refine_every = 100, continue_cull_post_densification: N changes every 100 steps for the whole run,
/root/reference freegaussian/freegaussian_model.py:404-436, :514-571).  bench.py measures a fixed-N steady state by
construction; this measures what a change of N costs the raster path: every quantity the host learns per shape (list
capacity, checkpoint-slot needs, even / long-segment / heavy-tile flags) is keyed by the tile grid and scaled by the
ratio of the Gaussian counts (ops.RasterContext.capacity_for), so the step after a refinement should run like the step
before it.

Usage: python scripts/refine_step_bench.py [layout] [n_gauss] [steps] [width] [height]
Prints one JSON object: per segment of constant N the mean / median step (GPU time between consecutive end-of-step
events), the series' mean over its steady state (median of each segment's second half), and the counters: stage-wise
fallbacks, list-capacity redos, full-size checkpoint allocations -- all counted AFTER the first four calls (the first call of a
process measures the shape; checkpoint-slot needs are read one call late)."""
import json
import os
import sys
import time

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import harness, ops  # noqa: E402
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import apply_layout, synthetic_scene  # noqa: E402

layout = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 350
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
H = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
split_frac = float(os.environ.get("REFINE_SPLIT_FRAC", "0.03"))  # share of the Gaussians a refinement splits / duplicates
DIAG = os.environ.get("REFINE_DIAG", "0") == "1"  # host timing and allocator counters around the first refinement (stderr)
dev = torch.device("cuda", 0)
N_VIEWS = 8
sc = apply_layout(synthetic_scene(n, W, H, n_views=N_VIEWS, sh_degree=3, seed=42), layout)
cfg = FreeGaussianModelConfig(background_color="random", num_downscales=0, warm_up=10**9, refine_every=100, refine_start=0)
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
cams = []
for v in range(N_VIEWS):
    c2w = torch.linalg.inv(sc.viewmats[v])
    c2w[:3, 1:3] *= -1
    K = sc.Ks[v]
    cams.append(Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, times=torch.tensor([[0.0]])))
gts = [torch.rand(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(v)) for v in range(N_VIEWS)]
# steps 3101 ...: SH degree 3; 3200, 3300, ... are refinement steps that densify (step % 3000 > num_train_data + refine_every)
# and cull (too-big culling is on beyond step 3000, screen-size culling below 4000)
step0 = 3101
ctx = ops.default_context


def set_split_threshold():
    """The reference's fixed densify_grad_thresh (0.0008) is tuned on trained scenes; on the synthetic scene with random
    targets it would split nearly everything.  Before a refinement step the threshold is set to the quantile of the
    accumulated statistic that splits / duplicates ``split_frac`` of the Gaussians -- the few percent per refinement a real
    run sees."""
    if model.xys_grad_norm is None:
        return
    avg = model.xys_grad_norm / model.vis_counts * 0.5 * float(max(model.last_size))
    k = max(int(avg.numel() * (1.0 - split_frac)), 1)
    model.config.densify_grad_thresh = float(avg.float().kthvalue(k).values)


def counters():
    return {"stagewise_raster_calls": ctx.stagewise_raster_calls, "capacity_redos": ctx.capacity_redos,
            "full_ckpt_allocs": ctx.full_ckpt_allocs, "long_calls": ctx.long_calls, "heavy_calls": ctx.heavy_calls}  # fmt: skip


# the process's first calls: the shape is measured (stage-wise), the slot needs arrive one call late
for i in range(4):
    harness.train_step(model, opts, cams[i % N_VIEWS], gts[i % N_VIEWS], step0 - 4 + i, metrics_every=10**9)
torch.cuda.synchronize()
base = counters()
events, counts, refined = [], [], []
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(steps):
    step = step0 + i
    if step % cfg.refine_every == 0:
        set_split_threshold()
    before = model.num_points
    if DIAG and 98 <= i <= 102:
        torch.cuda.synchronize()
        st0 = torch.cuda.memory_stats()
        t_host = time.perf_counter()
    harness.train_step(model, opts, cams[i % N_VIEWS], gts[i % N_VIEWS], step, num_train_data=N_VIEWS, metrics_every=10**9)
    if DIAG and 98 <= i <= 102:
        t_issue = time.perf_counter() - t_host
        torch.cuda.synchronize()
        st1 = torch.cuda.memory_stats()
        print(f"diag step {step}: host issue {1e3 * t_issue:.2f} ms, to completion {1e3 * (time.perf_counter() - t_host):.2f} ms, "
              f"device mallocs {st1['num_device_alloc'] - st0['num_device_alloc']}, frees {st1['num_device_free'] - st0['num_device_free']}, "
              f"segments {st1['segment.all.current']}, reserved {st1['reserved_bytes.all.current'] / 1e6:.0f} MB, N {model.num_points}", file=sys.stderr)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    events.append(e)
    counts.append(before)
    refined.append(model.num_points != before or step % cfg.refine_every == 0)
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b in zip([e0] + events[:-1], events)]
end = counters()
# segments of constant N (a refinement step belongs to the segment it ends: its own time holds the refinement kernels)
segments, start = [], 0
for i in range(steps):
    if refined[i] or i == steps - 1:
        segments.append((start, i + 1))
        start = i + 1
out_segments, total_plain, total_steady, total_calm, stalls = [], 0.0, 0.0, 0.0, []
for a, b in segments:
    seg = ms[a:b]
    plain = [t for t, r in zip(seg, refined[a:b]) if not r]  # without the refinement step itself
    if len(plain) < 8:
        continue
    half = sorted(plain[len(plain) // 2 :])
    steady = half[len(half) // 2]
    mean_plain = sum(plain) / len(plain)
    srt = sorted(plain)
    out_segments.append({"steps": [step0 + a, step0 + b - 1], "n_gauss": counts[a], "mean_ms": round(mean_plain, 4),
                         "median_ms": round(srt[len(srt) // 2], 4), "p90_ms": round(srt[int(0.9 * len(srt))], 4),
                         "steps_over_1.5x_steady": sum(t > 1.5 * steady for t in plain),
                         "steady_ms": round(steady, 4), "first8_ms": [round(t, 3) for t in plain[:8]],
                         "refinement_step_ms": round(seg[-1], 3) if refined[b - 1] else None})  # fmt: skip
    total_plain += sum(plain)
    total_steady += steady * len(plain)
    # (this platform's sporadic host stalls -- multiples of 15.6 ms, DESIGN.md section 4 -- land anywhere: listed, and a second
    # figure without them)
    for k, t in enumerate(plain):
        if t > 5.0 * steady:
            stalls.append({"step": step0 + a + k, "ms": round(t, 2)})
    total_calm += sum(t if t <= 5.0 * steady else steady for t in plain)
print(json.dumps({"layout": layout, "size": [n, W, H], "steps": steps, "split_frac": split_frac, "segments": out_segments,
                  "series_mean_over_steady_state": round(total_plain / max(total_steady, 1e-9), 4),
                  "steps_over_5x_steady": stalls,
                  "series_mean_over_steady_state_without_those": round(total_calm / max(total_steady, 1e-9), 4),
                  "counters_after_the_first_calls": {k: end[k] - base[k] for k in end},
                  "counters_of_the_first_calls": base,
                  "note": "train_step = step_cb, get_outputs, L1+SSIM loss, backward, six Adam groups, after_train_iter; "
                          "GPU time between end-of-step events; refinement steps excluded from the means (their kernels are "
                          "the densification's) and listed per segment"}))  # fmt: skip
