"""Every collective call site of the view-DP code through RCCL -- on the ONE GPU a box has.

``FG_DP_FORCE_COLLECTIVES=1`` makes ``viewdp._collective_world`` take the ``world > 1`` code paths on a 1-rank ``nccl``
process group: ``all_gather_into_tensor(async_op=True)`` issued from the autograd thread inside the backward, the head
all-reduce in slices from a side stream behind per-launch events (``_coalescing_manager``), the model path's flat
all-reduce, the sparse blocks' header read + a forced overflow (the dense repeat), ``all_reduce_densify_stats``,
``shared_seed``, ``all_reduce_model_grads``.  What a 1-rank communicator exercises: ProcessGroupNCCL's stream ordering,
work handles, the tensors' lifetimes across streams, the coalescing manager.  What it does NOT: the ring kernels, the
links, and -- an in-place all-reduce over one rank being the identity -- the ORDER of the head all-reduce against the
kernels that write its buffer (the out-of-place all-gathers would show an ordering mistake: their output would be stale).

Checked: over 20 steps with the view changing every step, the gradients of the forced path against the same step without
collectives (toggled per step in this process), at the repeatability bound of the backward itself (atomic accumulation
order: 1e-5 relative L2, tests/test_gpu_parity.py::test_backward_is_repeatable_within_tolerance) -- once on the default
stream, once on a non-default ``torch.cuda.Stream``.  Prints one JSON object; exit code 1 on a mismatch."""
import copy
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from freegaussian_amd import harness, rasterization, viewdp  # noqa: E402
from freegaussian_amd.model import FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from train_e2e import camera_from_viewmat  # noqa: E402

TOL = 1e-5
STEPS = int(os.environ.get("FG_ONE_RANK_STEPS", "20"))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def forced(on: bool):
    os.environ["FG_DP_FORCE_COLLECTIVES"] = "1" if on else "0"


def flat_params_leg(dev, exchange):
    """bench.py's step: FlatGaussianParams + factored (sliced head all-reduce, all-gather from inside the backward) or plain."""
    n, W, H = 200_000, 960, 540
    sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=42)
    A, B = viewdp.FlatGaussianParams.from_scene(sc, dev), viewdp.FlatGaussianParams.from_scene(sc, dev)
    vms, Ks = sc.viewmats.to(dev), sc.Ks.to(dev)
    vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    worst = 0.0

    def step(P, v):
        with (P.factored_exchange() if exchange == "factored" else P.direct_grads()):
            r, _, _ = rasterization(*P.raster_inputs(), vms[v : v + 1], Ks[v : v + 1], W, H, sh_degree=3, render_mode="RGB", packed=False, absgrad=True)
            r.backward(vr)
        if exchange == "plain":
            P.all_reduce_grads()

    for s in range(STEPS):
        v = s % 8
        forced(True)
        step(A, v)
        forced(False)
        step(B, v)
        torch.cuda.current_stream().synchronize()
        for k in A.params:
            worst = max(worst, rel_l2(A.params[k].grad, B.params[k].grad))
    return worst


def model_leg(dev, sparse, step0):
    """harness.train_step(dp=ModelViewDP): dense / sparse blocks (one forced overflow) / per-view means (deformation net on)."""
    n, W, H = 60_000, 480, 270
    sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=7)
    cfg = FreeGaussianModelConfig(background_color="white", num_downscales=0, warm_up=3000)
    torch.manual_seed(0)
    base = FreeGaussianModel(cfg, seed_points=sc.means, init_scales=-4.0)
    with torch.no_grad():
        gp = base.gauss_params
        gp["scales"].copy_(sc.scales.log()), gp["quats"].copy_(sc.quats)
        gp["opacities"].copy_(torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6))[:, None])
        gp["features_dc"].copy_(sc.colors[:, 0]), gp["features_rest"].copy_(sc.colors[:, 1:])
        for p in base.deform.parameters():
            p.mul_(0.3)
    models = [copy.deepcopy(base).to(dev).train() for _ in range(2)]
    opts = [harness.build_optimizers(m) for m in models]
    dps = [viewdp.ModelViewDP(m, sparse=sparse) for m in models]
    cams = [camera_from_viewmat(sc.viewmats[v], sc.Ks[v], W, H, v / 8) for v in range(8)]
    gts = [torch.rand(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(v)) for v in range(8)]
    worst, info = 0.0, {}
    for s in range(STEPS):
        v = (3 * s) % 8
        if sparse == "always" and s == 7:
            dps[0].force_overflow_next = dps[1].force_overflow_next = True
        grads = []
        for i, on in ((0, True), (1, False)):
            forced(on)
            m, o, dp = models[i], opts[i], dps[i]
            m.step_cb(step0 + s)
            for x in o.values():
                x.zero_grad(set_to_none=True)
            with dp.step():
                out = m.get_outputs(cams[v])
                ld = m.get_loss_dict(out, {"image": gts[v]})
                (ld["main_loss"] + ld["scale_reg"]).backward()
            torch.cuda.current_stream().synchronize()
            grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        # the two replicas take the SAME update (the forced one's) so that they stay comparable step after step
        for k, p in models[1].named_parameters():
            if k in grads[0]:
                p.grad.copy_(grads[0][k])
        for i in (0, 1):
            harness.apply_schedules(opts[i], step0 + s)
            from freegaussian_amd.optim import step_all

            step_all(opts[i].values())
        gd = [torch.cat([g[k].flatten() for k in sorted(g) if k.startswith("deform")]) for g in grads] if any(k.startswith("deform") for k in grads[0]) else None
        for k in grads[0]:
            if not k.startswith(("deform", "control")):
                worst = max(worst, rel_l2(grads[0][k], grads[1][k]))
        if gd is not None:
            worst = max(worst, rel_l2(gd[0], gd[1]))
    info = {"sparse_steps": dps[0].sparse_steps, "dense_steps": dps[0].dense_steps, "overflows": dps[0].sparse_overflows,
            "payload_form_last": dps[0].bytes_last_step.get("payload_form")}
    # the remaining call sites
    forced(True)
    models[0].after_train_iter(models[0].step)
    before = [models[0].xys_grad_norm.clone(), models[0].vis_counts.clone(), models[0].max_2Dsize.clone()]
    viewdp.sync_densify_stats(models[0])  # all_reduce_densify_stats (sum, max) + shared_seed (broadcast)
    after = [models[0].xys_grad_norm, models[0].vis_counts, models[0].max_2Dsize]
    info["densify_stats_identity"] = all(torch.equal(a, b) for a, b in zip(before, after))
    g0 = {k: p.grad.clone() for k, p in models[0].named_parameters() if p.grad is not None}
    viewdp.all_reduce_model_grads(models[0])
    info["all_reduce_model_grads_identity"] = all(torch.equal(g0[k], p.grad) for k, p in models[0].named_parameters() if k in g0)
    forced(False)
    return worst, info


def run_all(dev):
    res = {}
    for ex in ("factored", "plain"):
        res[f"flat_{ex}"] = flat_params_leg(dev, ex)
    for name, sparse, step0 in (("model_dense", "never", 100), ("model_sparse", "always", 100), ("model_per_view_means", "auto", 3100)):
        w, info = model_leg(dev, sparse, step0)
        res[name] = w
        res[name + "_info"] = info
    return res


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t0 = time.perf_counter()
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "steps": STEPS, "tolerance": TOL,
           "head_slices": int(os.environ.get("FG_DP_HEAD_SLICES", "4"))}
    out["default_stream"] = run_all(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out["side_stream"] = run_all(dev)
    side.synchronize()
    out["seconds"] = round(time.perf_counter() - t0, 1)
    bad = [(leg, k, v) for leg in ("default_stream", "side_stream") for k, v in out[leg].items()
           if (isinstance(v, float) and not v < TOL) or (isinstance(v, dict) and not all(x for x in v.values() if isinstance(x, bool)))]
    out["ok"] = not bad
    out["mismatches"] = bad
    import ctypes

    ctypes.CDLL(None).fflush(None)  # (RCCL's version banner sits in the C stdio buffer: out with it before the result line)
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
