#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the Gaussian raster path, forward + backward, on the
synthetic 1M-Gaussian 1920x1080 scene of BASELINE.json (configs "DyNeRF multi-view 1080p,
~1M Gaussians"; generator: freegaussian_amd/scenes.py north_star_scene).

One process per GPU.  Step s of rank r renders camera view (r + s) mod 8 of the shared scene (the
8-view ring of cfg4, all poses resident on the device) through ``freegaussian_amd.rasterization``
-- the drop-in for the call at reference freegaussian_model.py:847 -- and back-propagates a fixed
N(0,1) image gradient.  For N > 1 the Gaussian-parameter gradient (59 floats per Gaussian) is
exchanged over RCCL inside the timed region, as view-sharded training does every step.

Launching N ranks: either the caller does (``python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) or, when
``--gpus N`` > 1 is given WITHOUT ``WORLD_SIZE``, this process starts the N rank processes itself
-- before it has imported the package or made any GPU call -- relays rank 0's JSON line and
exits with the worst rank's code.

A step = one view per rank, forward + backward (+ exchange).  value = total pixels rendered by
all ranks / max-over-ranks wall time.  Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0) with the contract keys plus "roofline" (dominant kernel) and
"cpu_baseline" (the CPU oracle timed on a bounded crop of the same view, N=1 only)."""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch and the package (which maps libfgraster.so) are imported inside the functions that need
# them: the launcher branch of main() must start its rank processes before anything in this
# process can have touched the GPU.

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")  # newest first
N_VIEWS = 8


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-s", type=float, default=1.0,
                    help="seconds of untimed steps right before the timed region, with no host synchronisation in "
                    "between: the device's clocks follow its recent load (the dominant kernel ran 0.39 ms right "
                    "after an idle stretch and 0.31-0.34 ms 48 steps later), so a short timed region behind a pause "
                    "measures the ramp")
    ap.add_argument("--n-gauss", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--graph-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--stage-events", choices=("all", "dominant", "none"), default="dominant",
                    help="HIP events in the timed region: around every C-ABI call, only around the dominant kernel "
                         "(the stage table then comes from an untimed pass before it), or none")
    ap.add_argument("--fixed-view", action="store_true", help="render view `rank` every step instead of cycling the ring")
    ap.add_argument("--layout", type=str, default="uniform",
                    help="uniform (the headline's U([-2,2]^3) cloud) | clustered:<frac>:<extent> -- that fraction of the Gaussians "
                         "pulled into a ball of that extent at the centre (captured-scene-like: lists of thousands of entries, "
                         "saturated and unsaturated; scripts/clustered_check.py) | needles:<frac>:<ratio> -- that fraction made "
                         "anisotropic (one axis x ratio, one / 3: the needles and plates densification leaves) | parts joined "
                         "by + (freegaussian_amd.scenes.apply_layout).  The default run also measures two clustered and two "
                         "needle layouts in child processes and reports them under `clustered_layouts`; the headline stays the "
                         "uniform scene.")
    ap.add_argument("--no-clustered", action="store_true", help="skip the clustered-layout child runs")
    ap.add_argument("--cpu-crop", type=str, default="480x272")  # ~4 s of oracle time per run on the GPU box
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = pick the faster of 8 and 32 host threads")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launcher: --gpus N without WORLD_SIZE


def rank_environments(n, port, base=None):
    """The environment of each of the n rank processes (torch.distributed's env:// contract)."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))  # fmt: skip
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on these hosts
        e.setdefault("OMP_NUM_THREADS", "8")
        envs.append(e)
    return envs


def free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs a rank of this launch would see, counted WITHOUT importing torch or calling into HIP (either would
    open the driver in the launcher): the KFD topology nodes that have SIMDs, capped by the *_VISIBLE_DEVICES
    lists.  No /sys/class/kfd = no AMD GPU driver = 0."""
    import re

    base = "/sys/class/kfd/kfd/topology/nodes"
    n = 0
    try:
        for d in os.listdir(base):
            try:
                m = re.search(r"^simd_count\s+(\d+)", open(os.path.join(base, d, "properties")).read(), re.M)
            except OSError:
                continue
            n += bool(m and int(m.group(1)) > 0)
    except OSError:
        return 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch_ranks(args, argv):
    """Start args.gpus rank processes of this script, relay rank 0's stdout, return the exit code.
    Runs before this process has imported torch / freegaussian_amd or touched the GPU (devices are counted
    from sysfs); the children are fresh interpreters, never an exec of this one.  Every rank checks again
    for itself (LOCAL_RANK against its own device count) before it joins the process group."""
    import subprocess

    n = args.gpus
    backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not os.environ.get("FG_BENCH_ECHO"):
        have = visible_gpu_count()
        if have < n:
            print(f"[bench] --gpus {n} but only {have} GPU(s) visible: one rank per GPU over RCCL, devices are "
                  "never shared (FG_BENCH_BACKEND=gloo exercises the N-rank control flow on fewer GPUs)",
                  file=sys.stderr)  # fmt: skip
            return 2
    envs = rank_environments(n, free_port())
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r, e in enumerate(envs):
        # rank 0's stdout is the result line; the other ranks' stdout joins stderr
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()  # the exact process started above
            codes.append(p.wait())
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"[bench] rank exit codes (rank, code): {bad}", file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------


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


def _pmc_record(stage, workload_key):
    """The committed rocprofv3 PMC passes of this same command (profiles/rNN_pmc_traffic.json,
    produced by scripts/gpu_pmc.sh: separate FETCH_SIZE and WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md prescribes).  PMC collection cannot run inside the timed process, so these
    are recorded figures; None when absent or recorded for another workload."""
    kern = STAGE_KERNEL.get(stage)
    if kern is None:
        return None
    for fname in PMC_FILES:
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        rec = json.load(open(path))
        if rec.get("workload_key") != workload_key:
            continue
        for name, v in rec["kernels"].items():
            if name.startswith(kern):  # str.startswith takes the tuple of candidate kernel names
                return dict(v, source=f"profiles/{fname}")
    return None


def pmc_traffic(stage, workload_key):
    rec = _pmc_record(stage, workload_key)
    return None if rec is None else rec["hbm_bytes_per_launch"]


def pmc_valu(stage, workload_key):
    """Vector wave-instructions per launch of `stage`'s kernel (SQ_INSTS_VALU of the same PMC file)."""
    rec = _pmc_record(stage, workload_key)
    return None if rec is None else rec.get("valu_wave_instr_per_launch")


def isa_class_mix(stage):
    """Instruction-class shares of `stage`'s kernel (profiles/rNN_isa_class_mix.json, scripts/isa_class_mix.py), or None."""
    for fname in ("r05_isa_class_mix.json", "r04_isa_class_mix.json"):
        path = os.path.join(ROOT, "profiles", fname)
        if os.path.exists(path):
            rec = json.load(open(path))
            k = rec["kernels"].get(stage)
            if k is not None:
                return dict(k, cost_clocks=rec["cost_clocks"], source_file=f"profiles/{fname}")
    return None


def _rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _oracle_run(scene, view, K, cw, ch, sh_degree, vr, compositor=None, absgrad=False):
    import torch

    from oracle import raster_oracle as O

    names = ("means", "quats", "scales", "opacities", "colors")
    ins = [getattr(scene, n).clone().requires_grad_(True) for n in names]
    t0 = time.perf_counter()
    r, a, info = O.rasterization(*ins, scene.viewmats[view : view + 1], K[None], cw, ch, sh_degree=sh_degree,
                                 render_mode="RGB", packed=False, absgrad=absgrad, compositor=compositor)  # fmt: skip
    if absgrad:
        info["means2d"].retain_grad()
    (r * vr).sum().backward()
    return time.perf_counter() - t0, r.detach(), ins, info


def _raster_lists_are_subsequences(ginfo, info_ref, n_gauss):
    """The lists the compositing WALKED (``raster_flatten_ids`` / ``raster_isect_offsets``: binned from the
    footprint rectangles) against the oracle's radius-box lists: every (tile, Gaussian) pair must be one of
    the oracle's and the order inside every tile must be the oracle's -- an order-preserving subsequence."""
    import torch

    ids_f, offs_f = info_ref["flatten_ids"].long(), info_ref["isect_offsets"].reshape(-1).long()
    ids_r, offs_r = ginfo["raster_flatten_ids"].cpu().long(), ginfo["raster_isect_offsets"].cpu().reshape(-1).long()
    n_tiles = offs_f.numel() - 1
    tiles = torch.arange(n_tiles)
    key_f = torch.repeat_interleave(tiles, torch.diff(offs_f)) * n_gauss + ids_f
    key_r = torch.repeat_interleave(tiles, torch.diff(offs_r[: n_tiles + 1])) * n_gauss + ids_r[: int(offs_r[n_tiles])]
    srt, perm = torch.sort(key_f)
    at = torch.searchsorted(srt, key_r).clamp_max(max(srt.numel() - 1, 0))
    found = bool((srt[at] == key_r).all()) if key_r.numel() else True
    pos = perm[at]  # position of every walked entry in the oracle's list
    ordered = bool((pos[1:] > pos[:-1]).all()) if pos.numel() > 1 else True
    return found and ordered, int(key_r.numel()), int(key_f.numel())


def _hip_parity(scene, view, K, cw, ch, sh_degree, vr, r_ref, ins_ref, info_ref):
    """The HIP path on the same inputs: PSNR of the render, relative L2 of every gradient
    (parameters, screen-space means2d, absgrad) against the oracle's."""
    import math

    import torch

    from freegaussian_amd import rasterization

    names = ("means", "quats", "scales", "opacities", "colors")
    dev = torch.device("cuda", torch.cuda.current_device())
    gin = [getattr(scene, n).to(dev).requires_grad_(True) for n in names]
    rg, _, ginfo = rasterization(*gin, scene.viewmats[view : view + 1].to(dev), K[None].to(dev), cw, ch,
                                 sh_degree=sh_degree, render_mode="RGB", packed=False, absgrad=True)  # fmt: skip
    ginfo["means2d"].retain_grad()
    (rg * vr.to(dev)).sum().backward()
    mse = float(((rg.detach().cpu().double() - r_ref.double()) ** 2).mean())
    rel = {n: _rel_l2(g.grad.cpu(), o.grad) for n, g, o in zip(names, gin, ins_ref)}
    if getattr(info_ref["means2d"], "absgrad", None) is not None:  # oracle runs with absgrad=True only
        rel["means2d"] = _rel_l2(ginfo["means2d"].grad.cpu(), info_ref["means2d"].grad)
        rel["absgrad"] = _rel_l2(ginfo["means2d"].absgrad.cpu(), info_ref["means2d"].absgrad)
    sub_ok, n_walked, n_ref = _raster_lists_are_subsequences(ginfo, info_ref, scene.means.shape[0])
    return {
        "psnr_hip_vs_oracle_db": 200.0 if mse == 0 else min(200.0, -10.0 * math.log10(mse)),
        # info["flatten_ids"] is the reference's radius-box list (rebuilt on demand); the compositing walks the
        # footprint lists info["raster_flatten_ids"], which info["last_ids"] indexes
        "reference_lists_bit_exact": bool(torch.equal(ginfo["flatten_ids"].cpu(), info_ref["flatten_ids"])),
        "raster_lists_order_preserving_subsequence_of_reference": sub_ok,
        "raster_list_entries": n_walked,
        "reference_list_entries": n_ref,
        "grad_rel_l2_hip_vs_oracle": rel,
        "grad_rel_l2_hip_vs_oracle_max": max(rel.values()),
    }


def cpu_baseline(scene, view, crop, sh_degree, threads=0):
    """The pure-PyTorch CPU oracle (oracle/raster_oracle.py, a port: the reference's own raster is
    the absent CUDA-only gsplat) on a centre crop of the same view, all Gaussians projected; fwd +
    bwd.  Protocol of BASELINE.md section 2: 1 warm-up + 3 timed runs, median; the thread count is
    the faster of 8 and 32 in the warm-ups (the oracle is thousands of small tensor ops: every one
    is a fork/join over the thread pool, and on the 256-thread GPU box a full-width pool made a run
    take minutes instead of seconds), both warm-up times reported next to the host's thread count.
    The same crop is rendered and differentiated by the HIP path: PSNR and the relative L2 error of
    every gradient against the oracle's are part of the line."""
    import torch

    cw, ch = (int(x) for x in crop.split("x"))
    K = scene.Ks[view].clone()
    x0, y0 = (scene.width - cw) // 2, (scene.height - ch) // 2
    K[0, 2] -= x0
    K[1, 2] -= y0
    host = os.cpu_count() or 1
    vr = torch.randn(1, ch, cw, 3, generator=torch.Generator().manual_seed(1))
    candidates = [threads] if threads > 0 else sorted({min(host, 8), min(host, 32)})
    warm = {}
    for c in candidates:
        torch.set_num_threads(c)
        warm[c], r_ref, ins_ref, info_ref = _oracle_run(scene, view, K, cw, ch, sh_degree, vr)
    cores = min(warm, key=warm.get)
    torch.set_num_threads(cores)
    timed = [_oracle_run(scene, view, K, cw, ch, sh_degree, vr)[0] for _ in range(3)]
    dt = sorted(timed)[1]
    parity = _hip_parity(scene, view, K, cw, ch, sh_degree, vr, r_ref, ins_ref, info_ref) if torch.cuda.is_available() else {}
    return {
        "value": cw * ch / dt / 1e6,
        **parity,
        "grad_note": "the oracle's backward is the reference's order (T rebuilt from T_final = 1 - alpha_out walking back; "
        "oracle/raster_oracle.py header), the same as the C compositor's and the HIP kernels': one bar, 1e-4",
        "unit": "Mpix/s",
        "cores": cores,
        "host_cpus": host,
        "kind": "port",
        "runs_s": [round(x, 2) for x in timed],
        "warmup_s_by_threads": {str(k): round(v, 2) for k, v in warm.items()},
        "sample": f"centre {cw}x{ch} crop of view {view} of the same scene (all {scene.means.shape[0]} Gaussians "
        f"projected, {info_ref['flatten_ids'].numel()} tile intersections), fwd+bwd, 1 warm-up per thread count + 3 "
        f"timed runs, median {dt:.1f} s on {cores} of {host} host threads; pure-PyTorch CPU oracle (extrapolates by "
        "pixel count)",
    }


def cpu_full_frame(scene, view, sh_degree):
    """The WHOLE headline frame on the host: torch projection / SH / sort of the oracle with the
    scalar C compositing (oracle/fg_oracle.c, OpenMP over tiles) -- no crop, no extrapolation -- and
    the HIP path's render and gradients of the whole frame against it.  One run (~10-30 s)."""
    import torch

    from oracle import c_oracle as CO

    host = os.cpu_count() or 1
    torch.set_num_threads(min(host, 32))  # the torch stages (projection, SH, sort); the C compositing is OpenMP over all cores
    W, H = scene.width, scene.height
    K = scene.Ks[view]
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1))
    dt, r_ref, ins_ref, info_ref = _oracle_run(scene, view, K, W, H, sh_degree, vr, compositor=CO.composite, absgrad=True)
    parity = _hip_parity(scene, view, K, W, H, sh_degree, vr, r_ref, ins_ref, info_ref) if torch.cuda.is_available() else {}
    return {"value": W * H / dt / 1e6, "unit": "Mpix/s", "cores": host, "kind": "port", "seconds": round(dt, 2), **parity,
            "sample": f"the whole {W}x{H} frame of view {view}, fwd+bwd, 1 run; torch projection / SH / sort "
            f"({min(host, 32)} threads) + C compositing (forward OpenMP over all {host} host threads; backward one "
            "thread with double accumulators, so that its sums do not depend on scheduling)"}  # fmt: skip


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


def trained_scene_for(spec):
    """``trained:<file>``: a ``trained_scene.npz`` of scripts/train_e2e.py.  ``trained:auto`` (or ``trained``): the file
    ``data/trained_scene_r06.npz`` when it is there (the round's reference-schedule run, deformation net included; 46 MB, not
    in the history), else the scene is trained HERE, in this process, before anything is timed: scripts/train_e2e.py's run from
    random_init on the reference's schedule for 7000 steps with the deformation net left off (``warm_up`` beyond the run: the
    target is static, and the net -- 25 of every 27 ms of a step behind step 3000 -- is a dense MLP outside the raster path;
    ~15 s instead of ~110).  -> (Scene, provenance dict)."""
    from freegaussian_amd.scenes import load_trained_scene

    path = spec.split(":", 1)[1] if ":" in spec else "auto"
    if path == "auto":
        cand = os.path.join(ROOT, "data", "trained_scene_r06.npz")
        if os.path.exists(cand) and not os.environ.get("FG_BENCH_TRAIN_HERE"):
            path = cand
    if path != "auto":
        return load_trained_scene(path), {"source": os.path.relpath(path, ROOT)}
    import tempfile

    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import train_e2e as E

    out_dir = tempfile.mkdtemp(prefix="fg_trained_")
    t0 = time.perf_counter()
    # (FG_BENCH_TRAIN="steps,n_target,width,height,num_random": a smaller run -- the gate's check of this fallback, tests/test_e2e.py)
    steps, n_target, tw, th, n_rand = (int(x) for x in os.environ.get("FG_BENCH_TRAIN", "7000,200000,1920,1080,50000").split(","))
    _, rep, _ = E.train(steps=steps, n_target=n_target, width=tw, height=th, num_random=n_rand, warm_up=10**9, eval_at=(steps,),
                        out_dir=out_dir, log=lambda *a: None, save_checkpoint=False)
    prov = {"source": f"trained in this process: scripts/train_e2e.py, {steps} steps from random_init, reference schedule, deformation net off",
            "train_seconds": round(time.perf_counter() - t0, 1), "heldout_psnr_db": rep["evals"][-1]["heldout_psnr"],
            "N_final": rep["N_final"], "train_step_gpu_ms": rep["step_time_gpu_ms"], "policy_counters_end": rep["policy_counters"]["end"]}  # fmt: skip
    return load_trained_scene(os.path.join(out_dir, "trained_scene.npz")), prov


def make_scene(args):
    """The scene `--layout` names (+ provenance for the trained one); sets args.n_gauss / width / height from a trained file."""
    from freegaussian_amd.scenes import apply_layout, synthetic_scene

    if args.layout.split(":")[0] == "trained":
        scene, prov = trained_scene_for(args.layout)
        args.n_gauss, args.width, args.height, args.sh_degree = scene.means.shape[0], scene.width, scene.height, scene.sh_degree
        return scene, prov
    scene = synthetic_scene(args.n_gauss, args.width, args.height, n_views=N_VIEWS, sh_degree=args.sh_degree, seed=42)
    try:
        apply_layout(scene, args.layout)
    except ValueError as e:
        raise SystemExit(f"--layout: {e}")
    return scene, None


def graph_only(args):
    """Child-process leg: the step replayed as one hipGraph; prints one JSON object."""
    import torch

    from freegaussian_amd.graphed import GraphedRaster
    from freegaussian_amd.viewdp import FlatGaussianParams

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    scene, _ = make_scene(args)
    W, H = scene.width, scene.height
    params = FlatGaussianParams.from_scene(scene, dev)
    vms, Ks = scene.viewmats.to(dev), scene.Ks.to(dev)
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    gr = GraphedRaster(params, W, H, sh_degree=args.sh_degree)
    for s in range(N_VIEWS + 2):  # every view once: the list capacity settles on the largest
        v = 0 if args.fixed_view else s % N_VIEWS
        gr.step(vms[v : v + 1], Ks[v : v + 1], vr)
    # (Replays enqueued back to back WITHOUT the per-step flag read -- flags OR-ed on the device -- run slower on this
    # stack, 1.06-1.9 ms per step against 0.81: a second launch of a graph whose previous launch is still running
    # takes a slow path in the runtime.  The per-step read is also what a trainer needs before its optimizer step.)
    for s in range(max(8, int(args.settle_s * 1e3))):  # the device warm in front of the timed region, like the eager loop
        v = 0 if args.fixed_view else s % N_VIEWS
        gr.step(vms[v : v + 1], Ks[v : v + 1], vr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    overflows = 0
    for s in range(args.steps):
        v = 0 if args.fixed_view else s % N_VIEWS
        overflows += int(gr.step(vms[v : v + 1], Ks[v : v + 1], vr)[2])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"ms_per_step": dt / args.steps * 1e3, "value": args.steps * W * H / dt / 1e6, "unit": "Mpix/s",
                      "list_capacity": gr.capacity, "overflows": overflows,
                      "note": "fwd+bwd replayed as one hipGraph; the overflow flag is read back every step (the device "
                              "idles between replays)"}))  # fmt: skip


# The auxiliary child runs (clustered / needle layouts, the graphed replay) come BEFORE the headline measurement -- alone on
# the device, see clustered_children -- so they share one bounded budget: a hung or slow child must not eat the caller's
# outer timeout before the headline line is printed (each child at most 120 s, all of them FG_BENCH_CHILD_BUDGET_S, 300 s).
_CHILD_T0 = [None]


def child_budget_left() -> float:
    if _CHILD_T0[0] is None:
        _CHILD_T0[0] = time.perf_counter()
    return float(os.environ.get("FG_BENCH_CHILD_BUDGET_S", "300")) - (time.perf_counter() - _CHILD_T0[0])


def clustered_children(args):
    """The same measurement on two clustered layouts, each in a process of its own (the headline stays the uniform
    scene): what the path does on captured-scene-like content -- long lists, saturated and not -- and which machinery
    ran.  Run BEFORE this process touches the GPU: with the parent's (idle) queues on the device beside the child's, the
    second timed step of a child stalled 16-23 ms on some boxes (multiples of 15.6 ms, never in a process alone on the
    device)."""
    import subprocess

    res_all = {}
    for lay in ("clustered:0.5:0.4", "clustered:0.8:0.2", "needles:0.3:10", "clustered:0.5:0.4+needles:0.3:10", "trained:auto"):
        left = child_budget_left()
        if left < 20:
            res_all[lay] = {"error": "skipped: the children's shared time budget is spent (FG_BENCH_CHILD_BUDGET_S)"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--layout", lay, "--steps", "40", "--warmup", "10", "--settle-s",
               str(max(args.settle_s, 1.0)), "--no-cpu-baseline", "--no-graph", "--n-gauss", str(args.n_gauss), "--width", str(args.width),
               "--height", str(args.height), "--sh-degree", str(args.sh_degree)]  # fmt: skip
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=min(120, left))
            c = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1])
            # (ms_per_step: the child's wall-clock mean over its 40 timed steps; the median host time per step beside it)
            res_all[lay] = {"mpix_per_s": c["value"], "ms_per_step": c["ms_per_step"],
                            "ms_per_step_median": (c.get("host_step_ms") or {}).get("median"), "stage_ms": c["stage_ms"],
                            "I_raster": c["config"]["I_raster"], "longest_tile_list": c["config"].get("longest_tile_list"),
                            "long_segment_calls": c["config"].get("long_segment_calls"),
                            "heavy_tile_steps": c["config"].get("heavy_tile_steps"), "host_step_ms": c.get("host_step_ms"),
                            "list_capacity_redos_in_timed_region": c["config"].get("list_capacity_redos_in_timed_region"),
                            "path_events_in_timed_region": c.get("path_events_in_timed_region")}  # fmt: skip
            for k in ("N", "V", "I", "scene", "scene_statistics"):  # (what the trained layout is: its own N, cameras, shapes)
                if lay.startswith("trained") and k in c["config"]:
                    res_all[lay][k] = c["config"][k]
            res_all[lay]["host_step_ms_p99"] = (c.get("host_step_ms") or {}).get("p99")
        except Exception as e:
            res_all[lay] = {"error": repr(e)[:200]}
    return res_all


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.graph_only:
        return graph_only(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    if os.environ.get("FG_BENCH_ECHO"):
        # launcher self-test (tests/test_bench_launch.py): report the rank environment, touch nothing else
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                         | {"n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "gpus_arg": args.gpus}), flush=True)  # fmt: skip
        return

    clustered = None
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.layout == "uniform" and not args.no_clustered and not under_profiler()):
        import torch  # (importing torch and counting devices does not initialise the GPU)

        if torch.cuda.device_count() > 0:
            clustered = clustered_children(args)  # (before this process initialises the GPU)
    graphed = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_graph and not under_profiler():
        # the same step captured in one hipGraph (graphed.GraphedRaster), measured in a CHILD process (a failed capture
        # aborts the process; the headline line must survive), alone on the device like the children above.
        # Informational: the headline is the eager path, whose kernels can be timed individually.
        import subprocess

        import torch

        if torch.cuda.device_count() > 0:
            cmd = [sys.executable, os.path.abspath(__file__), "--graph-only", "--steps", str(args.steps), "--n-gauss",
                   str(args.n_gauss), "--width", str(args.width), "--height", str(args.height), "--sh-degree",
                   str(args.sh_degree), "--layout", args.layout] + (["--fixed-view"] if args.fixed_view else [])  # fmt: skip
            try:
                left = child_budget_left()
                if left < 20:
                    raise TimeoutError("skipped: the children's shared time budget is spent (FG_BENCH_CHILD_BUDGET_S)")
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=min(180, left))
                graphed = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1])
            except Exception as e:
                graphed = {"error": repr(e)[:200]}

    import datetime

    import torch
    import torch.distributed as dist

    from freegaussian_amd import ops, rasterization
    from freegaussian_amd.viewdp import FlatGaussianParams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is used", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the raster path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {ndev} GPU(s) visible; RCCL ranks never share a device")
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    force = world == 1 and os.environ.get("FG_DP_FORCE_COLLECTIVES") == "1"
    if force:
        # every collective call site of the step on ONE rank (viewdp._collective_world): a 1-rank communicator exercises
        # ProcessGroupNCCL's stream / work-handle / lifetime semantics on the one GPU of a box, not the ring kernels
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force:
        # RCCL ("nccl") over xGMI is the product path; FG_BENCH_BACKEND=gloo only exists so that the
        # N>1 control flow can be exercised on a 1-GPU box (ranks then share the device).  A bounded
        # timeout turns a desynchronised collective into an error instead of a hung box.
        tmo = datetime.timedelta(seconds=int(os.environ.get("FG_BENCH_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    scene, scene_prov = make_scene(args)
    W, H = scene.width, scene.height
    params = FlatGaussianParams.from_scene(scene, dev)  # flat parameter + flat gradient buffers
    vms, Ks = scene.viewmats.to(dev), scene.Ks.to(dev)  # all 8 poses resident
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)

    exchange = os.environ.get("FG_EXCHANGE", "factored") if (world > 1 or force) else "none"
    marks = []  # per step: HIP events at start / after forward / after backward / after the exchange
    counter = [0]
    last_info = [None]

    def step(timed=False):
        view = rank % N_VIEWS if args.fixed_view else (rank + counter[0]) % N_VIEWS
        counter[0] += 1
        vm, K = vms[view : view + 1], Ks[view : view + 1]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if timed else None
        if ev:
            ev[0].record()
        # gradients land directly in the flat gradient buffer (dense overwrite: no zeroing needed);
        # N>1: factored exchange (all-gather of the colour gradient + 44 B/Gaussian all-reduce + local
        # SH rebuild, viewdp.factored_exchange) unless FG_EXCHANGE=plain (one 236 B/Gaussian all-reduce)
        factored = exchange == "factored"
        with (params.factored_exchange() if factored else params.direct_grads()):
            means, quats, scales, opac, colors = params.raster_inputs()
            r, a, info = rasterization(means, quats, scales, opac, colors, vm, K, W, H, sh_degree=args.sh_degree,
                                       render_mode="RGB", packed=False, absgrad=True)  # fmt: skip
            if ev:
                ev[1].record()
            r.backward(vr)  # upstream dL/d render = fixed N(0,1) image
            if ev:
                ev[2].record()
        if exchange == "plain":
            params.all_reduce_grads()
        if ev:
            ev[3].record()
            marks.append(ev)
        # (EVERY loop of this file keeps the previous step's `info` alive while the next step runs -- the timed loop reads it
        # afterwards --, so the workspace pool sees the same two-buffer pattern from the first warm-up step on.  Round 5's
        # loops dropped it everywhere but in the timed region: the region's SECOND step then found the pool's only buffer
        # still referenced and asked the device for a new one -- the 4.6 / 15.7 ms steps of the clustered lines
        # (`path_events_in_timed_region.pool_new_buffers` = 1).)
        last_info[0] = info
        return info, view

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fallback_note = None
    if exchange == "factored":
        # The factored exchange has only ever run on gloo and on one GPU.  Every rank tries one step;
        # the outcome is agreed on with a MIN all-reduce of a flag, so either all ranks keep it or
        # all fall back to the plain all-reduce together (a host-side failure is deterministic and
        # hits every rank before its first collective; a rank that dies inside a collective ends the
        # job through the process-group timeout).
        ok, err = 1, None
        try:
            step()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            ok, err = 0, repr(e)[:200]
        flag = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            fallback_note = f"factored exchange failed on at least one rank ({err}); all ranks use the plain all-reduce"
            if rank == 0:
                print(f"[bench] {fallback_note}", file=sys.stderr)
            exchange = "plain"
    # warm-up: at least one pass over the whole ring, so that the speculative list capacity has seen every view
    for _ in range(max(args.warmup, 0)):
        step()
    fence()
    # Everything from here to the timed region keeps the device BUSY: its clocks follow its recent load with a time
    # constant of tens of milliseconds -- behind an idle stretch (a collection, an event summary) the dominant
    # kernel ran 0.39 ms and was still getting faster 48 steps later (0.31-0.34 ms; FG_BENCH_SERIES=1 prints the
    # series), so a 20-step region behind a pause timed the ramp, not the path.  Order: collection, calibration,
    # settle (--settle-s seconds of steps, no host synchronisation inside), stage pass, a shorter second settle behind
    # the stage pass's summary, fence, timed region.  Step counts are the same on every rank (rank 0 decides).
    gc.collect()
    gc.disable()  # a generation-2 collection inside a 50 ms timed region is a 1-3 ms outlier, not a property of the path
    settle_steps = 0

    def settle(seconds, per_step):
        n = torch.tensor([max(8, int(seconds / per_step))], device=dev)
        if world > 1:
            dist.broadcast(n, 0)
        for _ in range(int(n.item())):
            step(timed=args.stage_events == "all")
        return int(n.item())

    per_step = 1e-3
    if args.settle_s > 0:
        fence()
        t_cal = time.perf_counter()
        for _ in range(8):
            step()
        fence()
        per_step = max((time.perf_counter() - t_cal) / 8, 1e-5)
        settle_steps = 8 + settle(args.settle_s, per_step)
    # Stage pass (untimed): HIP events around EVERY C-ABI call and around forward / backward / exchange of
    # every step.  Sixteen event records per step cost ~0.06 ms of a ~1 ms step (measured: 1.045 ms with
    # them, 0.987 without), so the timed region below carries events around the dominant kernel only --
    # the two the roofline's duration needs -- and the stage table comes from this pass over the same views.
    stage_steps = min(args.steps, 24) if args.stage_events != "none" else 0
    ops.default_context.stage_timer = ops.StageTimer()
    t_pass = time.perf_counter()  # (the device is busy with the settle steps: this interval includes their tail)
    for _ in range(stage_steps):
        step(timed=True)
    fence()
    stage_pass_ms = (time.perf_counter() - t_pass) / max(stage_steps, 1) * 1e3
    stages = ops.default_context.stage_timer.summary() if stage_steps else {}
    ops.default_context.stage_timer = None
    kernel_names = ("fg_raster_bwd", "fg_raster_fwd", "fg_raster_composite_bwd", "fg_raster_composite_fwd",
                    "fg_preprocess_fwd", "fg_preprocess_bwd", "fg_bin_prepare", "fg_bin_emit_sort_capacity")
    cand = {s: v for s, v in stages.items() if s in kernel_names}
    dominant = max(cand, key=lambda s: cand[s]) if cand else "fg_raster_bwd"
    only = {"all": None, "dominant": {dominant}, "none": set()}[args.stage_events]
    ops.default_context.stage_timer = ops.StageTimer(only=only)
    if args.settle_s > 0:
        settle_steps += settle(max(args.settle_s / 3, 0.2), per_step)
    redo0 = ops.default_context.capacity_redos

    def path_events():
        c = ops.default_context
        return {"list_capacity_redos": c.capacity_redos, "plan_changes": c.plan_changes, "pool_new_buffers": c.pool_new_buffers,
                "pool_fallback_calls": c.pool_fallback_calls, "stagewise_raster_calls": c.stagewise_raster_calls,
                "full_ckpt_allocs": c.full_ckpt_allocs, **{"plan_change_" + k: v for k, v in c.plan_change_reasons.items()}}

    if ops.default_context.stage_timer is not None:
        # (the settle steps' events are not part of the timed region's averages; the timed region's own event pairs exist
        # before it starts)
        ops.default_context.stage_timer = ops.StageTimer(only=only, prewarm=(args.steps + 2) * max(len(only), 1) if only else 0)
    fence()
    events0 = path_events()
    t0 = time.perf_counter()
    views_seen = []
    host_marks = [t0]
    for _ in range(args.steps):
        info, view = step(timed=args.stage_events == "all")
        views_seen.append(view)
        host_marks.append(time.perf_counter())
    fence()
    dt_local = time.perf_counter() - t0
    events1 = path_events()
    gc.enable()
    timed_stages = ops.default_context.stage_timer.summary()
    dom_series = [a.elapsed_time(b) for a, b in ops.default_context.stage_timer.events.get(dominant, [])]
    ops.default_context.stage_timer = None
    if os.environ.get("FG_BENCH_SERIES"):  # per-step durations of the dominant kernel and host marks (diagnosis)
        print("[bench] dominant kernel ms per timed step:", [round(x, 4) for x in dom_series], file=sys.stderr)
        print("[bench] host ms per timed step:", [round((b - a) * 1e3, 4) for a, b in zip(host_marks, host_marks[1:])], file=sys.stderr)
    if args.stage_events == "all":
        stages = timed_stages
    stages = dict(stages)
    stages.update(timed_stages)  # the dominant kernel's duration: from the timed region itself
    redos = ops.default_context.capacity_redos - redo0
    t = torch.tensor([dt_local], device=dev, dtype=torch.float64)
    per_rank = [t.clone() for _ in range(world)]
    rank_devices = None
    if world > 1:
        dist.all_gather(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # which physical device every rank ran on (a SCALE record then shows that RCCL saw `world` distinct GPUs)
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_device": dev.index, "uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": getattr(props, "pci_bus_id", None), "name": props.name}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
    dt = float(t.item())

    N = args.n_gauss
    scene_stats = None
    if scene_prov is not None:
        from freegaussian_amd.scenes import scene_statistics

        scene_stats = scene_statistics(scene, info["radii"])
    V = int((info["radii"] > 0).sum())
    I_raster = int(info["raster_flatten_ids"].numel())  # entries the compositing walks (footprint rectangles)
    I = int(info["flatten_ids"].numel())  # the reference's list length (radius boxes): SURVEY section 8d's I
    P = W * H
    T = info["tile_width"] * info["tile_height"]
    k = (args.sh_degree + 1) ** 2
    nbits = 32 + max(T - 1, 1).bit_length()
    p = (nbits + 7) // 8  # passes the reference-style 64-bit-key sort would need (SURVEY.md §8d formula)
    tile_passes = (max(T - 1, 1).bit_length() + 7) // 8
    alg = algorithmic_bytes(N, V, I, P, T, k, p)
    wkey = f"{N}x{W}x{H}xsh{args.sh_degree}"
    # the dominant KERNEL: exchange / host-composed stages are not kernels of the path
    kernel_stages = {s: v for s, v in stages.items() if s in alg}
    no_events = not kernel_stages
    if no_events:  # --stage-events none: an A/B of the events' own cost, no roofline to report
        stages = dict(stages)
        kernel_stages = {"fg_raster_bwd": 1.0}
        stages["fg_raster_bwd"] = 1.0  # placeholder (ms); the roofline block is dropped below
    dom = max(kernel_stages, key=lambda s: kernel_stages[s])
    roof = {
        # achieved / peak / frac are the contract's HBM figures (no dense contraction, no MFMA);
        # `bound` names the pipe the counters show actually limits the kernel: "valu" for the raster kernels
        # (their measured HBM traffic is below the algorithmic bytes), "hbm" for the per-Gaussian passes
        "bound": "hbm",
        "kernel": dom,
        "achieved": alg.get(dom, 0) / (stages[dom] * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "traffic": pmc_traffic(dom, wkey),
        "algorithmic_bytes": alg.get(dom, 0),
        "avg_ms": stages[dom],
        "launch_counts": "I, V of the last timed step's view; duration averaged over the views of the timed region",
    }
    roof["frac"] = roof["achieved"] / roof["peak"]
    if dom in ("fg_raster_bwd", "fg_raster_fwd"):
        # the same figure on the bytes of the lists this library actually walks (I_raster entries: the footprint
        # rectangles drop the reference's dead (splat, tile) pairs) -- the contract's `frac` prices the reference's I
        walked = algorithmic_bytes(N, V, I_raster, P, T, k, p)[dom]
        roof["algorithmic_bytes_on_walked_lists"] = walked
        roof["frac_on_walked_lists"] = walked / (stages[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS
    if no_events:
        roof = {"bound": "hbm", "kernel": dom, "note": "--stage-events none: no kernel was timed"}
        stages = {}
    roof["hbm_frac"] = roof["frac"]
    if dom in ("fg_raster_bwd", "fg_raster_fwd") and not no_events:
        roof["bound"] = "valu"
        roof["frac_is"] = "algorithmic HBM bytes / duration / 8 TB/s (the contract's figure); valu_frac is beside it"
    if dom in ("fg_raster_bwd", "fg_raster_fwd"):
        roof["measured_limiter"] = ("vector issue + dependent-issue latency (VALU), not HBM: measured traffic is below "
                                    "the algorithmic bytes (L2 / Infinity Cache hits) -- see vector_issue_roofline")  # fmt: skip
    # Vector issue of the raster kernels (profiles/r01_valu_issue_rates.md): a SIMD issues one
    # full-rate wave64 fp32 instruction per 2 clocks (256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1229 G/s) but
    # only with ~4 ready wavefronts of dependent code; half-rate instructions (compare, select,
    # min/max, DPP) cost 4 clocks, exp / rcp / permlane swaps 8.  `frac` is against the 2-clock peak.
    issue = {}
    for st_name in ("fg_raster_bwd", "fg_raster_fwd"):
        vi = pmc_valu(st_name, wkey)
        if vi is not None and st_name in stages:
            peak = 1024 * 2.4e9 / 2  # wave-instructions per second, full-rate instructions
            rate = vi / (stages[st_name] * 1e-3)
            issue[st_name] = {"valu_wave_instr": vi, "avg_ms": stages[st_name],
                              "achieved_G_instr_per_s": rate / 1e9, "peak_G_instr_per_s": peak / 1e9,
                              "frac": rate / peak, "clocks_per_instr_per_simd": 1024 * 2.4e9 / rate}  # fmt: skip
            mix = isa_class_mix(st_name)
            if mix is not None:
                # the cost-weighted floor: the counter's instruction total x the kernel's class mix x the measured cost
                # of each class (profiles/r01_valu_issue_rates.md), spread over the chip's 1024 SIMDs at 2.4 GHz
                floor_ms = vi * mix["weighted_clocks_per_valu_instr"] / (1024 * 2.4e9) * 1e3
                issue[st_name]["cost_weighted_floor"] = {
                    "floor_ms": floor_ms, "floor_over_measured": floor_ms / stages[st_name],
                    "weighted_clocks_per_instr": mix["weighted_clocks_per_valu_instr"],
                    "class_shares": mix["shares"], "class_cost_clocks": mix["cost_clocks"],
                    "class_counts_static": {c: mix[c] for c in ("full", "half", "quarter")},
                    "note": "class shares = every vector instruction of the kernel's ISA counted once (static, "
                            "scripts/isa_class_mix.py -> " + mix["source_file"] + "); the instruction TOTAL is the hardware counter's",
                }
    if dom in issue and not no_events:
        roof["valu_frac"] = issue[dom]["frac"]
        roof["valu_wave_instr_per_launch"] = issue[dom]["valu_wave_instr"]
    # the HBM-bound stages next to it: measured traffic (same PMC file) over their HIP-event time
    hbm_stages = {}
    for st_name in ("fg_preprocess_fwd", "fg_preprocess_bwd", "fg_raster_fwd", "fg_raster_bwd"):
        tr = pmc_traffic(st_name, wkey)
        if tr is not None and st_name in stages:
            gbs = tr / (stages[st_name] * 1e-3) / 1e9
            hbm_stages[st_name] = {"traffic": tr, "algorithmic_bytes": alg[st_name], "avg_ms": stages[st_name],
                                   "traffic_over_algorithmic": tr / max(alg[st_name], 1),
                                   "achieved_GBs": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}  # fmt: skip
    survey_total = 280 * N + (176 + 24 * k) * V + (100 + 24 * p) * I + 56 * P + 8 * T
    needed = survey_total - 24 * p * I  # the 64-bit-key sort of the formula is not what this path runs

    # BASELINE.md section 3 protocol: per-step HIP-event times, median with p10 / p90
    def pct(xs, q):
        xs = sorted(xs)
        return xs[min(len(xs) - 1, max(0, int(round(q * (len(xs) - 1)))))]

    if not marks:  # --stage-events none: nothing was bracketed
        z = torch.cuda.Event(enable_timing=True)
        z.record()
        torch.cuda.synchronize()
        marks.append([z, z, z, z])
    t_fwd = [e[0].elapsed_time(e[1]) for e in marks]
    t_bwd = [e[1].elapsed_time(e[2]) for e in marks]
    t_fb = [a + b for a, b in zip(t_fwd, t_bwd)]
    t_xchg = [e[2].elapsed_time(e[3]) for e in marks]
    event_times = {
        "fwd_ms_median": pct(t_fwd, 0.5), "bwd_ms_median": pct(t_bwd, 0.5),
        "fwd_plus_bwd_ms": {"median": pct(t_fb, 0.5), "p10": pct(t_fb, 0.1), "p90": pct(t_fb, 0.9)},
        "mpix_per_s_per_gpu_from_median": P / max(pct(t_fb, 0.5) * 1e-3, 1e-9) / 1e6,
        "from": "timed region" if args.stage_events == "all" else f"stage pass of {stage_steps} untimed steps before the timed region",
    }  # fmt: skip
    step_s = dt / args.steps
    out = {
        "metric": "Mpixels/s fwd+bwd @ 1M Gaussians 1080p",
        "value": world * args.steps * P / dt / 1e6,
        "unit": "Mpix/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (f"north-star cfg4: {N} Gaussians" if scene_prov is None else f"TRAINED scene ({N} Gaussians, its own cameras)")
            + f", {W}x{H}, SH degree {args.sh_degree}, 1 view per rank per step, "
            + ("view = rank (fixed)" if args.fixed_view else f"view = (rank + step) mod {N_VIEWS} around the 8-view ring")
            + ", fwd+bwd, RGB, absgrad" + (f", {'RCCL' if backend == 'nccl' else backend} gradient exchange ({exchange})" if world > 1 else ""),
            "layout": args.layout,
            **({} if scene_prov is None else {"scene": scene_prov, "scene_statistics": scene_stats}),
            "N": N, "V": V, "I": I, "P": P, "T": T, "k": k,
            "I_raster": I_raster,
            "I_note": "I = tile intersections of the reference algorithm (radius-box rectangles; the formulas of SURVEY "
            "section 8d use it); I_raster = entries of the lists actually binned and composited: the footprint rectangles "
            "and, inside them, the footprint masks (the blocks of the rectangle the ellipse itself reaches) drop (splat, "
            "tile) pairs in which the splat reaches alpha >= 1/255 nowhere (same image, same gradients)",
            # (per image size, where they pay: the share of the rectangles' (splat, tile) pairs the masks kept in the shape's last
            # masked calls; above RasterContext.mask_keep_max the shape runs without them and looks again every 64th call)
            "footprint_masks": {"on_for_the_next_call": bool(ops.current().masks_on((dev, info["tile_width"], info["tile_height"]))),
                                "kept_share_of_last_masked_calls": [round(x, 3) for x in (ops.current().mask_keep.get((dev, info["tile_width"], info["tile_height"])) or [[]])[0]]},
            "counts_are_for_view": view,
            "binning": {
                "supertile": "supertile (csrc/stbin.hip): count by corner marks -> column scan -> one 8-byte element per "
                "(Gaussian, 2x2-tile supertile) scattered into the supertile's segment -> one LDS sort per supertile by "
                "(depth bits, id), four tile lists read off it; 6 launches, the raster job lists built inside the scatter launch",
                "depthfirst": f"depth-first: 4-pass 32-bit sort of N + {tile_passes}-pass tile sort of I; 26 launches",
            }[ops.default_context.binning] + f" (the 64-bit-key sort of the SURVEY formula would be {p} passes over I)",
            "parallelism": f"view-dp{world}",
            "list_capacity_redos_in_timed_region": redos,
            "longest_tile_list": int((info["raster_isect_offsets"].reshape(-1)[1:] - info["raster_isect_offsets"].reshape(-1)[:-1]).max()),
            "long_segment_calls": ops.default_context.long_calls,
            "heavy_tile_steps": ops.default_context.heavy_calls,
            # checkpoint slots (4352 B each) of the last step's buffer: the forward's states for the backward's list shares
            "seg_ckpt_slots": ops.default_context.last_seg_slots,
            "seg_ckpt_mb": round(ops.default_context.last_seg_slots * 4352 / 1e6, 1),
            "untimed_steps_before_timed_region": {"warmup": args.warmup, "stage_pass": stage_steps, "settle": settle_steps,
                                                  "order": "warm-up, settle (no host sync inside), stage pass, shorter settle, fence, timed region"},
        },
        "roofline": roof,
        "vector_issue_roofline": issue,
        "measured_hbm_traffic_by_stage": hbm_stages,
        "hip_event_times": event_times,
        # SURVEY section 8d's protocol figure (hipEvent pairs around fwd and bwd, median), beside the wall-clock `value`
        "hip_event_mpix_per_s": world * event_times["mpix_per_s_per_gpu_from_median"],
        "whole_step": {
            # headline: the bytes THIS path needs (SURVEY §8d total without the 24 p I term of a
            # 64-bit-key sort it does not run)
            "algorithmic_bytes": needed,
            "achieved_GBs": needed / step_s / 1e9,
            "frac_of_hbm_peak": needed / step_s / 1e9 / HBM_PEAK_GBS,
            "survey_formula_with_64bit_sort": {
                "algorithmic_bytes": survey_total,
                "achieved_GBs": survey_total / step_s / 1e9,
                "frac_of_hbm_peak": survey_total / step_s / 1e9 / HBM_PEAK_GBS,
                "note": f"includes 24*p*I = {24 * p * I} B of {p}-pass 64-bit-key sort traffic the depth-first binning avoids",
            },
        },
        # host-side time between consecutive returns of step(): the host waits for the GPU once per step (the
        # list length), so these follow the GPU's progress; a stall of the host or the driver shows up here
        "host_step_ms": (lambda d: {"median": sorted(d)[len(d) // 2], "p90": pct(d, 0.9), "p99": pct(d, 0.99), "max": max(d), "argmax": d.index(max(d)),
                                    "over_1.5x_median": sum(1 for x in d if x > 1.5 * sorted(d)[len(d) // 2])})(
            [(b - a) * 1e3 for a, b in zip(host_marks, host_marks[1:])]),
        # what the PATH itself did inside the timed region that could make one step slower than its neighbours: a list that
        # outgrew its speculative capacity (second binning), a one-call plan that differed from the shape's previous one (a
        # policy flip or a new capacity: new plan, maybe new buffers), a workspace the pool could not serve (a device
        # allocation), a stage-wise fallback.  All zero + a multi-millisecond host_step_ms.max = the stall is the host's.
        "path_events_in_timed_region": {k: events1.get(k, 0) - events0.get(k, 0) for k in events1},
        "stage_ms": {s: round(v, 4) for s, v in sorted(stages.items(), key=lambda kv: -kv[1])},
        "stage_pass_ms_per_step": round(stage_pass_ms, 4),
        "stage_ms_from": ("HIP events around every C-ABI call in the timed region" if args.stage_events == "all" else
                          f"{dominant}: HIP events in the timed region; the other stages: a pass of {stage_steps} steps over the "
                          "same views with events around every call (untimed; stage_pass_ms_per_step is that pass's own wall "
                          "time per step: sixteen event records per step cost a few percent of it)"),
    }  # fmt: skip
    if exchange != "none":
        xm = pct(t_xchg, 0.5)
        out["exchange"] = {
            "kind": exchange,
            "backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else " (host-staged; control-flow check only)"),
            "exchange_ms": xm,
            "exchange_ms_note": "median HIP-event time from the end of the backward to the end of the exchange on rank 0 "
            "(the factored all-gather is issued inside the backward and is partly hidden there)",
            "what": (f"factored: all-gather {12 * (N + 1)} B per rank (issued inside the backward), all-reduce {44 * N} B, "
                     f"local SH rebuild of {192 * N} B") if exchange == "factored"
            else f"plain: one all-reduce of {params.flat_grad.numel() * 4} B",
            "fallback": fallback_note,
        }  # fmt: skip
        out["exchange"]["head_all_reduce_slices"] = int(os.environ.get("FG_DP_HEAD_SLICES", "4")) if exchange == "factored" else 1
        out["exchange"]["forced_on_one_rank"] = bool(force)
    if world > 1:
        out["per_rank_mpix_per_s"] = [args.steps * P / float(x.item()) / 1e6 for x in per_rank]
        out["per_rank_device"] = rank_devices
        out["distinct_devices"] = len({(d or {}).get("uuid") or (d or {}).get("local_device") for d in rank_devices or []})
    if rank == 0:
        out["clocks_after_timed_region"] = gpu_clocks()
    if graphed is not None:
        out["graphed"] = graphed
    if clustered is not None:
        out["clustered_layouts"] = clustered
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, view, args.cpu_crop, args.sh_degree, args.cpu_threads)
            out["cpu_full_frame"] = cpu_full_frame(scene, view, args.sh_degree)
        if world > 1 or force:
            # RCCL writes a version banner through C stdio on its first collective; on a pipe that buffer is flushed when the
            # process exits -- BEHIND this line.  Flush it now: the result stays the last line of stdout.
            import ctypes

            ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
