"""View-sharded DP training of FreeGaussianModel, W ranks (torchrun): ranks render different views,
exchange gradients (viewdp.ModelViewDP: the factored exchange; FG_DP_EXCHANGE=plain: viewdp.all_reduce_model_grads) and densification statistics
(viewdp.sync_densify_stats) and must stay in lockstep -- same Gaussian count and bit-identical
parameters after a run that includes refinements.  On a 1-GPU box: FG_BENCH_BACKEND=gloo."""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import harness as Hn  # noqa: E402
from freegaussian_amd import viewdp  # noqa: E402
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
torch.cuda.set_device(dev)
backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

W, H, n = 160, 96, 4000
sc = synthetic_scene(n, W, H, n_views=8, sh_degree=3, seed=7)
torch.manual_seed(0)
# FG_DP_WARM_UP = n: the deformation MLP is active from step n on (default: never) -- every rank then renders its own
# deformed means (the factored exchange sends the view direction along) and the MLP gradients ride in the small all-reduce
cfg = FreeGaussianModelConfig(background_color="white", num_downscales=0, warm_up=int(os.environ.get("FG_DP_WARM_UP", 10**9)), refine_start=10,
                              refine_every=10, reset_alpha_every=30, densify_grad_thresh=1e-4, stop_screen_size_at=0,
                              sh_degree_interval=1)
model = FreeGaussianModel(cfg, seed_points=sc.means, init_scales=-4.0)
with torch.no_grad():
    gp = model.gauss_params
    gp["scales"].copy_(sc.scales.log())
    gp["quats"].copy_(sc.quats)
    gp["opacities"].copy_(torch.logit(sc.opacities.clamp(1e-4, 1 - 1e-4))[:, None])
    gp["features_dc"].copy_(sc.colors[:, 0])
    gp["features_rest"].copy_(sc.colors[:, 1:])
model = model.to(dev).train()


def camera(v):
    c2w = torch.linalg.inv(sc.viewmats[v])
    c2w[:3, 1:3] *= -1
    K = sc.Ks[v]
    return Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H,
                  times=torch.zeros(1, 1))


target = copy.deepcopy(model).eval()
with torch.no_grad():
    target.gauss_params["features_dc"].add_(0.25)
    gts = {v: target.get_outputs(camera(v))["rgb"].clamp(0, 1) for v in range(8)}
opts = Hn.build_optimizers(model)
hist = []


def grads_of_one_step(dp_obj, step):
    """Gradients of every parameter after one exchanged forward + backward of this rank's view (no optimizer step)."""
    import contextlib

    for p in model.parameters():
        p.grad = None
    model.step_cb(step)
    v = rank % 8
    with (dp_obj.step() if dp_obj is not None else contextlib.nullcontext()):
        out = model.get_outputs(camera(v))
        Hn.main_loss(out["rgb"], gts[v]).backward()
    if dp_obj is None:
        viewdp.all_reduce_model_grads(model)
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


# the two exchanges on the same model state: same averaged gradients up to fp32 summation order
probe_step = max(cfg.warm_up, 10)
dp_probe = viewdp.ModelViewDP(model)  # (FG_DP_SPARSE = auto | always | never: the gathered blocks' form)
g_f, g_p = grads_of_one_step(dp_probe, probe_step), grads_of_one_step(None, probe_step)
worst = max(float((g_f[k] - g_p[k]).norm() / g_p[k].norm().clamp_min(1e-30)) for k in g_p)
if rank == 0:
    print(f"factored vs plain exchange, worst relative L2 over {len(g_p)} gradients: {worst:.2e}")
assert set(g_f) == set(g_p) and worst < 1e-5, worst
# the same step again: the first one's counts have decided the blocks' form (sparse: rows of (id, g (, direction)) at 1.25 x
# the largest count) -- the same gradients (the expansion is exact; two backward passes differ by the order of their float
# atomics); then with a capacity that is far too small: every rank sees the overflow in the gathered headers and the gather
# is repeated densely


def same(a, b):
    return max(float((a[k] - b[k]).norm() / b[k].norm().clamp_min(1e-30)) for k in b) < 1e-5


g_2 = grads_of_one_step(dp_probe, probe_step)
assert same(g_2, g_p), "sparse blocks changed a gradient"
form_2 = dp_probe.bytes_last_step["payload_form"]
if dp_probe._sparse_plan is not None:
    dp_probe._sparse_plan = (8,) + dp_probe._sparse_plan[1:]
    g_3 = grads_of_one_step(dp_probe, probe_step)
    assert dp_probe.sparse_overflows == 1 and same(g_3, g_p), "overflow path"
if rank == 0:
    print(f"second probe step: {form_2} blocks, {dp_probe.bytes_last_step['rows_with_colour_gradient']} rows of "
          f"{dp_probe.bytes_last_step['gaussians']}; sparse steps {dp_probe.sparse_steps}, overflows {dp_probe.sparse_overflows}")
if os.environ.get("FG_DP_SPARSE") == "always":
    assert form_2 == "sparse" and dp_probe.sparse_steps >= 1
for p in model.parameters():
    p.grad = None
# FG_DP_EXCHANGE = factored (default: viewdp.ModelViewDP) | plain (one all-reduce of every gradient)
exchange = os.environ.get("FG_DP_EXCHANGE", "factored")
dp = viewdp.ModelViewDP(model) if exchange == "factored" else None
for i in range(25):
    v = (i * world + rank) % 8  # this rank's view of the step
    hist.append(Hn.train_step(model, opts, camera(v), gts[v], 10 + i, num_train_data=2, dp=dp,
                              grad_sync=None if dp is not None else viewdp.all_reduce_model_grads,
                              stats_sync=viewdp.sync_densify_stats))
if rank == 0 and dp is not None:
    print("exchange bytes per rank in the last step:", dp.bytes_last_step)
    print(f"steps with sparse / dense blocks: {dp.sparse_steps} / {dp.dense_steps}, overflows {dp.sparse_overflows}")
torch.cuda.synchronize()
sig = torch.cat([torch.tensor([float(model.num_points)], device=dev)] +
                [p.detach().double().sum().float().reshape(1) for p in model.parameters()])
sigs = [torch.empty_like(sig) for _ in range(world)]
dist.all_gather(sigs, sig)
same = all(torch.equal(s, sigs[0]) for s in sigs)
counts = [h["gaussian_count"] for h in hist]
if rank == 0:
    print(f"world={world} gaussian counts {counts[0]} -> {counts[-1]} ({len(set(counts))} distinct), "
          f"loss {hist[0]['loss']:.4f} -> {hist[-1]['loss']:.4f}")
    print("dp lockstep ok" if same and len(set(counts)) > 1 else "dp lockstep MISMATCH")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if same else 1)
