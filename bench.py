#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the Gaussian raster path, forward + backward, on the
synthetic 1M-Gaussian 1920x1080 scene of BASELINE.json (configs "DyNeRF multi-view 1080p,
~1M Gaussians"; generator: freegaussian_amd/scenes.py north_star_scene).

One process per GPU; each rank renders its own camera view of the shared scene (view = rank
mod 8) through ``freegaussian_amd.rasterization`` -- the drop-in for the call at reference
freegaussian_model.py:847 -- and back-propagates a fixed N(0,1) image gradient.  For N > 1 the
flat Gaussian-parameter gradient buffer (59 floats per Gaussian) is all-reduced over RCCL
inside the timed region, as view-sharded training does every step.

A step = one view per rank, forward + backward (+ all-reduce).  value = total pixels rendered
by all ranks / max-over-ranks wall time.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract keys plus "roofline" (dominant kernel) and
"cpu_baseline" (the CPU oracle timed on a bounded crop of the same view, N=1 only)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from freegaussian_amd.viewdp import FlatGaussianParams  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-gauss", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-crop", type=str, default="960x544")  # ~15 s of oracle time on the GPU box
    return ap.parse_args()


def algorithmic_bytes(N, V, I, P, T, k, p):
    """SURVEY.md §8d per-stage algorithmic HBM bytes for one view (fp32)."""
    return {
        "fg_preprocess_fwd": 44 * N + (12 * k + 44) * V,
        "fg_preprocess_bwd": (88 + 12 * k) * V + 44 * N + 12 * 16 * N,
        "fg_bin_prepare": 8 * N + 2 * 4 * 16 * N + 16 * N,
        "fg_bin_emit_sort": 8 * I + 2 * 16 * I + 4 * I,
        "fg_bin_emit_sort_capacity": 8 * I + 2 * 16 * I + 4 * I,
        "fg_project_fwd": 44 * N + 32 * V,
        "fg_sh_pack_fwd": 16 * N + (12 * k + 32) * V + 64 * N,
        "fg_sh_fwd": 12 * k * V + 12 * V,
        "fg_tile_bin": 12 * I,
        "fg_sort_pairs": 24 * p * I,
        "fg_tile_ranges": 8 * I + 8 * T,
        "fg_raster_fwd": 40 * I + 20 * P,
        "fg_raster_bwd": 40 * I + 36 * P + 44 * V,
        "fg_project_bwd": 88 * V + 44 * N,
        "fg_sh_bwd": 12 * k * V + 12 * 16 * N,
    }


# stage (C-ABI entry point) -> the single kernel it launches, for the PMC traffic lookup
STAGE_KERNEL = {
    "fg_raster_bwd": ("raster_bwd_kernel", "raster_bwd_mixed_kernel"),
    "fg_raster_fwd": ("raster_fwd_mixed_kernel", "raster_fwd_kernel"),
    "fg_preprocess_fwd": ("preprocess_fwd_kernel",),
    "fg_preprocess_bwd": ("preprocess_bwd_kernel",),
}


def pmc_traffic(stage, workload_key):
    """HBM bytes per launch of `stage`'s kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/r01_pmc_traffic.json, produced by scripts/gpu_pmc.sh: separate FETCH_SIZE and
    WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes).  PMC collection cannot run inside
    the timed process, so this is a recorded figure; None when the file is absent or was recorded
    for another workload."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    kern = STAGE_KERNEL.get(stage)
    if kern is None or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get("workload_key") != workload_key:
        return None
    for name, v in rec["kernels"].items():
        if name.startswith(kern):  # str.startswith takes the tuple of candidate kernel names
            return v["hbm_bytes_per_launch"]
    return None


def pmc_valu(stage, workload_key):
    """Vector wave-instructions per launch of `stage`'s kernel (SQ_INSTS_VALU of the same PMC file)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    kern = STAGE_KERNEL.get(stage)
    if kern is None or not os.path.exists(path):
        return None
    rec = json.load(open(path))
    if rec.get("workload_key") != workload_key:
        return None
    for name, v in rec["kernels"].items():
        if name.startswith(kern):
            return v.get("valu_wave_instr_per_launch")
    return None


def cpu_baseline(scene, view, crop, sh_degree):
    """The CPU oracle (oracle/raster_oracle.py, a port: the reference's own raster is the absent
    CUDA-only gsplat) on a centre crop of the same view, all Gaussians projected; fwd + bwd."""
    from oracle import raster_oracle as O

    cw, ch = (int(x) for x in crop.split("x"))
    K = scene.Ks[view].clone()
    x0, y0 = (scene.width - cw) // 2, (scene.height - ch) // 2
    K[0, 2] -= x0
    K[1, 2] -= y0
    # the oracle is thousands of small tensor ops: beyond a few threads they only add
    # synchronisation cost, so it is run on (and reported for) at most 8 threads
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    ins = [t.clone().requires_grad_(True) for t in (scene.means, scene.quats, scene.scales, scene.opacities, scene.colors)]
    g = torch.Generator().manual_seed(1)
    vr = torch.randn(1, ch, cw, 3, generator=g)
    t0 = time.perf_counter()
    r, a, info = O.rasterization(*ins, scene.viewmats[view : view + 1], K[None], cw, ch, sh_degree=sh_degree,
                                 render_mode="RGB", packed=False)  # fmt: skip
    (r * vr).sum().backward()
    dt = time.perf_counter() - t0
    # PSNR of the HIP render against the oracle render of the same crop
    psnr_db = None
    if torch.cuda.is_available():
        dev = torch.device("cuda", torch.cuda.current_device())
        with torch.no_grad():
            rg, _, _ = rasterization(*[t.detach().to(dev) for t in ins], scene.viewmats[view : view + 1].to(dev),
                                     K[None].to(dev), cw, ch, sh_degree=sh_degree, render_mode="RGB", packed=False)  # fmt: skip
        mse = float(((rg.cpu().double() - r.detach().double()) ** 2).mean())
        psnr_db = 200.0 if mse == 0 else min(200.0, -10.0 * __import__("math").log10(mse))
    return {
        "value": cw * ch / dt / 1e6,
        "psnr_hip_vs_oracle_db": psnr_db,
        "unit": "Mpix/s",
        "cores": cores,
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"centre {cw}x{ch} crop of view {view} of the same scene (all {scene.means.shape[0]} Gaussians "
        f"projected, {info['flatten_ids'].numel()} tile intersections), fwd+bwd, 1 run, {dt:.1f} s; "
        "pure-PyTorch CPU oracle (extrapolates by pixel count)",
    }


def under_profiler():
    """rocprofv3 preloads a library that initialises the GPU in every process it starts: child
    programs (rocm-smi, the graph leg) must then not be spawned -- the GPU boxes refuse that exec."""
    return "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)


def gpu_clocks():
    """Engine / memory clock levels as rocm-smi reports them right after the timed region (BASELINE.md
    section 3 asks for the clocks to be noted); a child process, best effort."""
    import subprocess

    if under_profiler():
        return {"skipped": "running under a profiler"}

    try:
        res = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--json"], capture_output=True, text=True, timeout=30)
        card = next(iter(json.loads(res.stdout).values()))
        return {k: v for k, v in card.items() if any(c in k.lower() for c in ("sclk", "mclk", "fclk"))}
    except Exception as e:
        return {"error": repr(e)[:120]}


def graph_only(args):
    """Child-process leg: the step replayed as one hipGraph; prints one JSON object."""
    from freegaussian_amd.graphed import GraphedRaster

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scene = synthetic_scene(args.n_gauss, args.width, args.height, n_views=8, sh_degree=args.sh_degree, seed=42)
    W, H = scene.width, scene.height
    params = FlatGaussianParams.from_scene(scene, dev)
    vm, K = scene.viewmats[:1].to(dev), scene.Ks[:1].to(dev)
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    gr = GraphedRaster(params, W, H, sh_degree=args.sh_degree)
    for _ in range(5):
        gr.step(vm, K, vr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    overflows = 0
    for _ in range(args.steps):
        overflows += int(gr.step(vm, K, vr)[2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"ms_per_step": dt / args.steps * 1e3, "value": args.steps * W * H / dt / 1e6, "unit": "Mpix/s",
                      "list_capacity": gr.capacity, "overflows": overflows,
                      "note": "fwd+bwd replayed as one hipGraph; the overflow flag is read back every step"}))  # fmt: skip


def main():
    args = parse()
    if args.graph_only:
        return graph_only(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the raster path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if world > 1:
        # RCCL ("nccl") over xGMI is the product path; FG_BENCH_BACKEND=gloo only exists so that the
        # N>1 control flow can be exercised on a 1-GPU box (ranks then share the device)
        backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            if local_rank >= ndev:
                raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPUs visible")
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    scene = synthetic_scene(args.n_gauss, args.width, args.height, n_views=8, sh_degree=args.sh_degree, seed=42)
    view = rank % 8
    W, H = scene.width, scene.height
    params = FlatGaussianParams.from_scene(scene, dev)  # flat parameter + flat gradient buffers
    vm = scene.viewmats[view : view + 1].to(dev)
    K = scene.Ks[view : view + 1].to(dev)
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)

    exchange = os.environ.get("FG_EXCHANGE", "factored")
    marks = []  # per step: HIP events at start / after forward / after backward / after the exchange

    def step(timed=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if ev:
            ev[0].record()
        # gradients land directly in the flat gradient buffer (dense overwrite: no zeroing needed);
        # N>1: factored exchange (all-gather of the colour gradient + 44 B/Gaussian all-reduce + local
        # SH rebuild, viewdp.factored_exchange) unless FG_EXCHANGE=plain (one 236 B/Gaussian all-reduce)
        factored = world > 1 and exchange == "factored"
        with (params.factored_exchange() if factored else params.direct_grads()):
            means, quats, scales, opac, colors = params.raster_inputs()
            r, a, info = rasterization(means, quats, scales, opac, colors, vm, K, W, H, sh_degree=args.sh_degree,
                                       render_mode="RGB", packed=False, absgrad=True)  # fmt: skip
            if ev:
                ev[1].record()
            r.backward(vr)  # upstream dL/d render = fixed N(0,1) image
            if ev:
                ev[2].record()
        if world > 1 and not factored:
            params.all_reduce_grads()
        if ev:
            ev[3].record()
            marks.append(ev)
        return info

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1 and exchange == "factored":
        # insurance for a path no 1-GPU box can exercise over RCCL: a host-side failure of the factored
        # exchange (deterministic, hence on every rank alike) falls back to the plain all-reduce
        try:
            info = step()
            fence()
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] factored exchange failed ({e!r}); falling back to FG_EXCHANGE=plain", file=sys.stderr)
            exchange = "plain"
    for _ in range(args.warmup):
        info = step()
    fence()
    ops.stage_timer = ops.StageTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = step(timed=True)
    fence()
    dt = time.perf_counter() - t0
    stages = ops.stage_timer.summary()
    ops.stage_timer = None
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    N = args.n_gauss
    V = int((info["radii"] > 0).sum())
    I = int(info["flatten_ids"].numel())
    P = W * H
    T = info["tile_width"] * info["tile_height"]
    k = (args.sh_degree + 1) ** 2
    nbits = 32 + max(T - 1, 1).bit_length()
    p = (nbits + 7) // 8  # passes the reference-style 64-bit-key sort would need (SURVEY.md §8d formula)
    tile_passes = (max(T - 1, 1).bit_length() + 7) // 8
    alg = algorithmic_bytes(N, V, I, P, T, k, p)
    dom = max(stages, key=lambda s: stages[s])
    roof = {
        "bound": "hbm",
        "kernel": dom,
        "achieved": alg.get(dom, 0) / (stages[dom] * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "traffic": pmc_traffic(dom, f"{N}x{W}x{H}xsh{args.sh_degree}"),
        "algorithmic_bytes": alg.get(dom, 0),
        "avg_ms": stages[dom],
    }
    roof["frac"] = roof["achieved"] / roof["peak"]
    # Vector issue of the raster kernels (profiles/r01_valu_issue_rates.md): a SIMD issues one
    # full-rate wave64 fp32 instruction per 2 clocks (256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1229 G/s) but
    # only with ~4 ready wavefronts of dependent code; half-rate instructions (compare, select,
    # min/max, DPP) cost 4 clocks, exp / rcp / permlane swaps 8.  `frac` is against the 2-clock peak.
    issue = {}
    for st_name in ("fg_raster_bwd", "fg_raster_fwd"):
        vi = pmc_valu(st_name, f"{N}x{W}x{H}xsh{args.sh_degree}")
        if vi is not None and st_name in stages:
            peak = 1024 * 2.4e9 / 2  # wave-instructions per second, full-rate instructions
            rate = vi / (stages[st_name] * 1e-3)
            issue[st_name] = {"valu_wave_instr": vi, "avg_ms": stages[st_name],
                              "achieved_G_instr_per_s": rate / 1e9, "peak_G_instr_per_s": peak / 1e9,
                              "frac": rate / peak, "clocks_per_instr_per_simd": 1024 * 2.4e9 / rate}  # fmt: skip
    # the HBM-bound stages next to it: measured traffic (same PMC file) over their HIP-event time
    hbm_stages = {}
    for st_name in ("fg_preprocess_fwd", "fg_preprocess_bwd", "fg_raster_fwd", "fg_raster_bwd"):
        tr = pmc_traffic(st_name, f"{N}x{W}x{H}xsh{args.sh_degree}")
        if tr is not None and st_name in stages:
            gbs = tr / (stages[st_name] * 1e-3) / 1e9
            hbm_stages[st_name] = {"traffic": tr, "avg_ms": stages[st_name], "achieved_GBs": gbs,
                                   "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}  # fmt: skip
    total_alg = 280 * N + (176 + 24 * k) * V + (100 + 24 * p) * I + 56 * P + 8 * T
    # BASELINE.md section 3 protocol: per-step HIP-event times, median with p10 / p90
    def pct(xs, q):
        xs = sorted(xs)
        return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]

    t_fwd = [e[0].elapsed_time(e[1]) for e in marks]
    t_bwd = [e[1].elapsed_time(e[2]) for e in marks]
    t_fb = [a + b for a, b in zip(t_fwd, t_bwd)]
    t_xchg = [e[2].elapsed_time(e[3]) for e in marks]
    event_times = {
        "fwd_ms_median": pct(t_fwd, 0.5), "bwd_ms_median": pct(t_bwd, 0.5),
        "fwd_plus_bwd_ms": {"median": pct(t_fb, 0.5), "p10": pct(t_fb, 0.1), "p90": pct(t_fb, 0.9)},
        "mpix_per_s_per_gpu_from_median": P / (pct(t_fb, 0.5) * 1e-3) / 1e6,
    }  # fmt: skip
    if world > 1:
        event_times["exchange_ms_median_after_backward"] = pct(t_xchg, 0.5)
        if exchange == "factored":
            event_times["exchange"] = (f"factored: all-gather {12 * (N + 1)} B per rank (issued inside the backward), "
                                       f"all-reduce {44 * N} B, local SH rebuild of {192 * N} B")  # fmt: skip
        else:
            event_times["exchange"] = f"plain: one all-reduce of {params.flat_grad.numel() * 4} B"
    out = {
        "metric": "Mpixels/s fwd+bwd @ 1M Gaussians 1080p",
        "value": world * args.steps * P / dt / 1e6,
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"north-star cfg4: {N} Gaussians, {W}x{H}, SH degree {args.sh_degree}, 1 view per rank "
            f"per step (8-view ring), fwd+bwd, RGB, absgrad" + (f", RCCL gradient exchange ({exchange})" if world > 1 else ""),
            "N": N, "V": V, "I": I, "P": P, "T": T, "k": k,
            "binning": f"depth-first: 4-pass 32-bit sort of N + {tile_passes}-pass tile sort of I "
            f"(the 64-bit-key sort of the SURVEY formula would be {p} passes over I)",
            "parallelism": f"view-dp{world}",
        },
        "roofline": roof,
        "vector_issue_roofline": issue,
        "measured_hbm_traffic_by_stage": hbm_stages,
        "hip_event_times": event_times,
        "whole_step": {
            "algorithmic_bytes": total_alg,
            "algorithmic_bytes_without_sort": total_alg - 24 * p * I,
            "achieved_GBs": total_alg / (dt / args.steps) / 1e9,
            "frac_of_hbm_peak": total_alg / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
        },
        "stage_ms": {s: round(v, 4) for s, v in sorted(stages.items(), key=lambda kv: -kv[1])},
    }  # fmt: skip
    if rank == 0:
        out["clocks_after_timed_region"] = gpu_clocks()
    if world == 1 and not args.no_graph and rank == 0 and not under_profiler():
        # the same step captured in one hipGraph (graphed.GraphedRaster), measured in a CHILD process
        # (a failed capture aborts the process; the headline line must survive).  Informational:
        # the headline above is the eager path, whose kernels can be timed individually.
        import subprocess

        cmd = [sys.executable, os.path.abspath(__file__), "--graph-only", "--steps", str(args.steps), "--n-gauss",
               str(args.n_gauss), "--width", str(args.width), "--height", str(args.height), "--sh-degree",
               str(args.sh_degree)]  # fmt: skip
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            out["graphed"] = json.loads(res.stdout.strip().splitlines()[-1])
        except Exception as e:
            out["graphed"] = {"error": repr(e)[:200]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, view, args.cpu_crop, args.sh_degree)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
