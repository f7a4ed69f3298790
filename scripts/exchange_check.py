"""Factored view-DP exchange vs the plain flat all-reduce, W ranks (torchrun), real kernels.
On a 1-GPU box: FG_BENCH_BACKEND=gloo and every rank shares the GPU.  Prints 'exchange ok' on rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from freegaussian_amd.viewdp import FlatGaussianParams  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
torch.cuda.set_device(dev)
backend = os.environ.get("FG_BENCH_BACKEND", "nccl")
dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
sc = synthetic_scene(60000, 320, 192, n_views=8, sh_degree=3, seed=42)
vm, K = sc.viewmats[rank % 8 : rank % 8 + 1].to(dev), sc.Ks[rank % 8 : rank % 8 + 1].to(dev)
vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(1)).to(dev)
per_view = os.environ.get("FG_PER_VIEW_MEANS") == "1"  # every rank renders its own displaced means
shift = 0.01 * (rank + 1) * torch.sin(torch.arange(sc.means.shape[0] * 3, device=dev).float()).view(-1, 3)


def render(p):
    means, quats, scales, opac, colors = p.raster_inputs()
    if per_view:
        means = means + shift  # stands in for the per-view deformation (:832-845)
    r, _, _ = rasterization(means, quats, scales, opac, colors, vm, K, sc.width, sc.height, sh_degree=3,
                            packed=False, absgrad=True)
    r.backward(vr)


grads = []
for mode in ("plain", "factored"):
    p = FlatGaussianParams.from_scene(sc, dev)
    p.flat_grad.fill_(float("nan"))
    if mode == "plain":
        if per_view:  # the displaced means are not a leaf: ordinary accumulation into the flat buffer
            p.zero_grad()
            render(p)
        else:
            with p.direct_grads():
                render(p)
        p.all_reduce_grads()
    else:
        if per_view:
            p.zero_grad()
        with p.factored_exchange(per_view_means=per_view):
            render(p)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(p.flat_grad).all()), mode
    assert all(q.grad is not None for q in p.raster_inputs()), mode
    grads.append(p.flat_grad.clone())
a, b = grads
n = sc.means.shape[0]
err_rest = float((a[: 11 * n] - b[: 11 * n]).norm() / a[: 11 * n].norm())
err_col = float((a[11 * n :] - b[11 * n :]).norm() / a[11 * n :].norm())
ok = err_rest < 1e-6 and err_col < 1e-5
t = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
if rank == 0:
    print(f"world={world} rel err non-colour {err_rest:.2e} colour {err_col:.2e}")
    print("exchange ok" if float(t) == 1.0 else "exchange MISMATCH")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if float(t) == 1.0 else 1)
