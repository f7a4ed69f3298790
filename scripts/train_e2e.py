"""End-to-end training of a scene on the reference's schedule, through the HIP path only.

What the reference's product IS (freegaussian_model.py:160-196, :404-571, :626-633, :827, :965-983; schedule
freegaussian_config.py:28-95; loop freegaussian_pipeline.py:53-66): a model from ``random_init`` (50 000 points in a cube
of side 10, scales from the three nearest neighbours) trained for thousands of steps, one full image per step, at 1/4 ->
1/2 -> full resolution (3000 / 6000), SH degree ``step // 1000``, densify / cull every 100 steps from 500, opacity reset
every 3000, screen-size culls to 4000, deformation net from ``warm_up`` = 3000, L1 + 0.2 (1 - SSIM), the optimizer table
of the method spec.  The data here: ``scenes.room_scene`` -- a hidden target of ~200k anisotropic Gaussians laid out as
surfaces (floor, walls, shells, rods), 32 training + 8 held-out cameras at mixed radii including poses among the content;
the ground truth is rendered ONCE with the HIP path (RGBA, kept as uint8 like the reference's ``cache_images_type``).

Reports held-out PSNR at the evaluation steps, N(t), the step-time series (GPU time between consecutive end-of-step
events: p50 / p99 per resolution phase), every launch-policy counter that moved, and writes

* ``<out>/step-%09d.ckpt``     the nerfstudio-layout checkpoint (freegaussian_amd.io)
* ``<out>/trained_scene.npz``  what the rasterizer is handed for eight of the scene's own cameras (activated parameters,
                               deformation applied at one time): ``bench.py --layout trained:<that file>``
* ``<out>/train_e2e.json``     the report

Usage: python scripts/train_e2e.py [--steps 7000] [--out gpurun_out/e2e] [--n-target 200000] [--width 1920 --height 1080]
                                   [--eval-at 1000,3000,7000] [--warm-up 3000] [--graphed] [--seed 42]"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def camera_from_viewmat(vm: torch.Tensor, K: torch.Tensor, W: int, H: int, t: float):
    """A ``model.Camera`` (OpenGL camera-to-world, as nerfstudio's) from an OpenCV world-to-camera matrix."""
    from freegaussian_amd.model import Camera

    c2w = torch.linalg.inv(vm)
    c2w[:3, 1:3] *= -1
    return Camera(c2w[None, :3].contiguous(), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H,
                  times=torch.tensor([[float(t)]]))  # fmt: skip


@torch.no_grad()
def render_ground_truth(scene, dev):
    """[V] uint8 RGBA images [H,W,4] of the hidden target through the HIP path: straight (un-premultiplied) colour and
    alpha, the layout of the reference's synthetic sets (composited over the step's background by ``get_gt_img`` /
    ``composite_with_background``, freegaussian_model.py:900-922)."""
    from freegaussian_amd import rasterization

    g = [getattr(scene, n).to(dev) for n in ("means", "quats", "scales", "opacities", "colors")]
    out = []
    for v in range(scene.viewmats.shape[0]):
        r, a, _ = rasterization(*g, scene.viewmats[v : v + 1].to(dev), scene.Ks[v : v + 1].to(dev), scene.width, scene.height,
                                sh_degree=scene.sh_degree, render_mode="RGB", packed=False)  # fmt: skip
        straight = torch.where(a > 1e-6, r / a.clamp_min(1e-6), torch.zeros_like(r)).clamp(0, 1)
        out.append((torch.cat([straight, a], -1)[0] * 255.0 + 0.5).to(torch.uint8))
    return out


@torch.no_grad()
def evaluate(model, cams, gts, idxs):
    """Mean per-image PSNR over the views ``idxs`` at full resolution, eval mode (RGB+ED, the model's fixed background)."""
    was = model.training
    model.eval()
    vals = []
    for i in idxs:
        out = model.get_outputs(cams[i])
        gt = model.composite_with_background(model.get_gt_img(gts[i]), out["background"])
        vals.append(float(-10.0 * torch.log10(torch.nn.functional.mse_loss(out["rgb"], gt))))
    model.train(was)
    return sum(vals) / len(vals), vals


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))] if xs else None


@torch.no_grad()
def export_scene(model, scene, view_ids, t: float, path: str):
    """The rasterizer's inputs for this model at time ``t`` -- activations and the deformation applied, exactly what
    ``get_outputs`` hands over (freegaussian_model.py:832-851) -- with ``view_ids`` of the scene's own cameras."""
    import numpy as np

    from freegaussian_amd.utils import transform_points

    means = model.gauss_params["means"]
    scales = torch.exp(model.gauss_params["scales"])
    quats = model.gauss_params["quats"] / model.gauss_params["quats"].norm(dim=-1, keepdim=True)
    if model.step >= model.config.warm_up:
        times = torch.full((means.shape[0], 1), float(t), device=means.device)
        d_xyz, d_rot, d_scale = model.deform(means, times)
        means, quats, scales = transform_points(d_xyz, means), quats + d_rot, scales + d_scale
    colors = torch.cat([model.gauss_params["features_dc"][:, None, :], model.gauss_params["features_rest"]], 1)
    np.savez(path, means=means.cpu().numpy(), quats=quats.cpu().numpy(), scales=scales.cpu().numpy(),
             opacities=torch.sigmoid(model.gauss_params["opacities"]).squeeze(-1).cpu().numpy(),
             colors=colors.cpu().numpy().astype("float16"), viewmats=scene.viewmats[view_ids].numpy(), Ks=scene.Ks[view_ids].numpy(),
             width=scene.width, height=scene.height, sh_degree=model.config.sh_degree, step=model.step)  # fmt: skip


def counters(ctx):
    return {k: getattr(ctx, k) for k in ("stagewise_raster_calls", "capacity_redos", "full_ckpt_allocs", "long_calls", "heavy_calls")}


def train(steps: int = 7000, n_target: int = 200_000, width: int = 1920, height: int = 1080, seed: int = 42,
          eval_at=(1000, 3000, 7000), warm_up: int = 3000, graphed: bool = False, out_dir=None, num_random: int = 50_000,
          log=print, config_overrides=None, device=None, save_checkpoint=True):  # fmt: skip
    from freegaussian_amd import harness, ops
    from freegaussian_amd.model import FreeGaussianModel, FreeGaussianModelConfig
    from freegaussian_amd.scenes import room_scene

    dev = torch.device("cuda", 0) if device is None else device
    torch.manual_seed(seed)
    scene, meta = room_scene(n_target, width, height, n_views=40, seed=seed)
    t0 = time.perf_counter()
    gts = render_ground_truth(scene, dev)
    torch.cuda.synchronize()
    log(f"[e2e] target: {scene.means.shape[0]} Gaussians, {len(gts)} views {width}x{height} rendered in {time.perf_counter() - t0:.1f} s")
    cams = [camera_from_viewmat(scene.viewmats[v], scene.Ks[v], width, height, meta["times"][v]) for v in range(len(gts))]
    train_ids, test_ids = meta["train"], meta["test"]
    cfg = FreeGaussianModelConfig(warm_up=warm_up, num_random=num_random, **(config_overrides or {}))  # everything else: the reference's defaults
    model = FreeGaussianModel(cfg).to(dev).train()  # random_init: cube of side 10, knn scales (:150-196)
    opts = harness.build_optimizers(model)
    gm = None
    if graphed:
        from freegaussian_amd.graphed import GraphedModelStep

        gm = GraphedModelStep(model)
    ctx = ops.default_context
    g = torch.Generator().manual_seed(seed)
    order: list = []
    marks, host, n_series, refinements, evals, phase_of = [], [], [], [], [], []
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    marks.append(ev0)
    c0 = counters(ctx)
    flags_by_phase = {}
    t_train = time.perf_counter()
    for step in range(1, steps + 1):
        if not order:  # the full-image data manager's order: a fresh permutation of the training views each pass
            order = [train_ids[i] for i in torch.randperm(len(train_ids), generator=g).tolist()]
        v = order.pop()
        n_before = model.num_points
        res = harness.train_step(model, opts, cams[v], gts[v], step, num_train_data=len(train_ids), graphed=gm, metrics_every=100)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)
        host.append(time.perf_counter())
        n_series.append(model.num_points)
        phase_of.append(model._get_downscale_factor())
        if model.num_points != n_before or (step % cfg.refine_every == 0 and step >= cfg.refine_start):
            refinements.append({"step": step, "before": n_before, "after": model.num_points})
        if "loss" in res and step % 500 == 0:
            log(f"[e2e] step {step}: loss {res['loss']:.4f} train-psnr {res['psnr']:.2f} N {model.num_points} "
                f"({time.perf_counter() - t_train:.1f} s)")  # fmt: skip
        if step in eval_at:
            torch.cuda.synchronize()
            te = time.perf_counter()
            p, per = evaluate(model, cams, gts, test_ids)
            ptrain, _ = evaluate(model, cams, gts, train_ids[:8])
            evals.append({"step": step, "heldout_psnr": p, "heldout_per_view": [round(x, 2) for x in per], "train_psnr_8_views": ptrain,
                          "N": model.num_points, "eval_s": round(time.perf_counter() - te, 2)})  # fmt: skip
            log(f"[e2e] step {step}: held-out PSNR {p:.2f} dB (train views {ptrain:.2f}), N {model.num_points}")
            flags_by_phase[step] = counters(ctx)
            # (the evaluation's own events must not be booked as a training step)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks[-1] = e
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t_train
    gpu_ms = [a.elapsed_time(b) for a, b in zip(marks, marks[1:])]
    by_phase = {}
    for d, ms, n in zip(phase_of, gpu_ms, n_series):
        by_phase.setdefault(d, []).append(ms)
    refine_steps = {r["step"] for r in refinements}
    series = {}
    for d, xs in by_phase.items():
        series[f"downscale_{d}"] = {"steps": len(xs), "mean_ms": sum(xs) / len(xs), "p50_ms": pct(xs, 0.5), "p90_ms": pct(xs, 0.9),
                                    "p99_ms": pct(xs, 0.99), "max_ms": max(xs)}  # fmt: skip
    plain = [ms for s, ms in enumerate(gpu_ms, 1) if s not in refine_steps and (s - 1) not in refine_steps]
    report = {
        "what": "scripts/train_e2e.py: FreeGaussianModel from random_init on the reference's schedule, HIP path only",
        "target": {"n_gauss": int(scene.means.shape[0]), "views_train": len(train_ids), "views_heldout": len(test_ids),
                   "size": [width, height], "parts": meta["parts"]},
        "config": {k: getattr(cfg, k) for k in ("warm_up", "refine_every", "resolution_schedule", "num_downscales", "sh_degree_interval",
                                                "sh_degree", "stop_split_at", "refine_start", "cull_alpha_thresh", "cull_scale_thresh",
                                                "reset_alpha_every", "densify_grad_thresh", "densify_size_thresh", "cull_screen_size",
                                                "split_screen_size", "stop_screen_size_at", "ssim_lambda", "background_color",
                                                "num_random", "random_scale")},  # fmt: skip
        "steps": steps, "graphed_low_res": bool(graphed), "train_seconds": round(train_s, 1),
        "evals": evals,
        "N_final": model.num_points,
        "N_at": {str(s): n_series[s - 1] for s in range(500, steps + 1, 500)},
        "refinements": refinements[:: max(1, len(refinements) // 40)],
        "step_time_gpu_ms": series,
        "step_time_gpu_ms_plain_steps": {"steps": len(plain), "p50": pct(plain, 0.5), "p99": pct(plain, 0.99), "max": max(plain) if plain else None,
                                         "what": "steps that neither refine nor follow a refinement"},  # fmt: skip
        "policy_counters": {"start": c0, "at_eval": flags_by_phase, "end": counters(ctx)},
        "learned_state": {"long_shapes": {str(k): v for k, v in ctx.long_shapes.items()}, "heavy_shapes": {str(k): v for k, v in ctx.heavy_shapes.items()},
                          "uneven_left": {str(k): v for k, v in ctx.uneven_left.items()}, "even_calls": {str(k): v for k, v in ctx.even_calls.items()}},  # fmt: skip
    }
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        from freegaussian_amd import io as fio

        if save_checkpoint:
            fio.save_checkpoint(out_dir, steps, model, opts)
        bench_views = train_ids[:: max(1, len(train_ids) // 8)][:8]
        export_scene(model, scene, bench_views, meta["times"][bench_views[0]], os.path.join(out_dir, "trained_scene.npz"))
        report["bench_views"] = bench_views
        json.dump(report, open(os.path.join(out_dir, "train_e2e.json"), "w"), indent=1)
    return model, report, (scene, meta, cams, gts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--out", type=str, default=os.path.join(ROOT, "gpurun_out", "e2e"))
    ap.add_argument("--n-target", type=int, default=200_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--eval-at", type=str, default="1000,3000,7000")
    ap.add_argument("--warm-up", type=int, default=3000)
    ap.add_argument("--graphed", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    ev = tuple(int(x) for x in a.eval_at.split(",") if x)
    _, report, _ = train(a.steps, a.n_target, a.width, a.height, a.seed, ev, a.warm_up, a.graphed, a.out)
    print(json.dumps({k: report[k] for k in ("steps", "train_seconds", "evals", "N_final", "step_time_gpu_ms", "policy_counters")}))


if __name__ == "__main__":
    main()
