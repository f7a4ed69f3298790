"""View-sharded DP host logic on CPU: flat buffers, and a world_size-2 gloo all-reduce that must
equal the single-process sum over both ranks' views (SURVEY.md §4 'multi-GPU without a cluster')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from freegaussian_amd.scenes import synthetic_scene
from freegaussian_amd.viewdp import (FLOATS_PER_GAUSSIAN, FlatGaussianParams, all_reduce_densify_stats,
                                     shared_seed, views_for_rank)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_flat_buffer_views_and_inplace_grad_accumulation():
    sc = synthetic_scene(100, 64, 64)
    fp = FlatGaussianParams.from_scene(sc, "cpu")
    assert fp.flat.numel() == 100 * FLOATS_PER_GAUSSIAN == sum(p.numel() for p in fp.params.values())
    means, quats, scales, opac, colors = fp.raster_inputs()
    assert torch.equal(means, sc.means) and torch.equal(colors, sc.colors)
    loss = (means * 2).sum() + (colors * 3).sum() + opac.sum()
    loss.backward()
    # gradients landed in the flat buffer, not in fresh tensors
    assert means.grad.data_ptr() == fp.flat_grad.data_ptr()
    assert float(fp.flat_grad[: 300].min()) == 2.0
    assert float(fp.flat_grad.sum()) == 2 * 300 + 3 * 4800 + 100
    fp.zero_grad()
    assert float(fp.flat_grad.abs().sum()) == 0.0
    assert means.grad.data_ptr() == fp.flat_grad.data_ptr()


def test_views_for_rank_partition():
    got = sorted(v for r in range(3) for v in views_for_rank(8, r, 3))
    assert got == list(range(8))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synthetic_scene(200, 64, 64, n_views=4)
    fp = FlatGaussianParams.from_scene(sc, "cpu")
    # a stand-in differentiable "render" per view (the raster itself needs a GPU): any function
    # of the shared parameters and the rank's views exercises the same exchange step
    for v in views_for_rank(4, rank, world):
        m, q, s, o, c = fp.raster_inputs()
        w = sc.viewmats[v][:3, :3]
        (((m @ w.T).sin() * o[:, None]).sum() + (c[:, 0] * s).sum() * (v + 1) + q.square().sum()).backward()
    fp.all_reduce_grads(average=False)
    g, vis, mx = torch.full((5,), float(rank + 1)), torch.full((5,), rank + 1), torch.full((5,), float(rank))
    all_reduce_densify_stats(g, vis, mx)
    seed = shared_seed(1234 + rank)
    r = torch.rand(3)
    if rank == 0:
        torch.save({"grad": fp.flat_grad.clone(), "g": g, "vis": vis, "mx": mx, "seed": seed, "r": r}, out)
    else:
        torch.save({"seed": seed, "r": r}, out + ".1")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_matches_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got, got1 = torch.load(out), torch.load(out + ".1")
    sc = synthetic_scene(200, 64, 64, n_views=4)
    fp = FlatGaussianParams.from_scene(sc, "cpu")
    for v in range(4):
        m, q, s, o, c = fp.raster_inputs()
        w = sc.viewmats[v][:3, :3]
        (((m @ w.T).sin() * o[:, None]).sum() + (c[:, 0] * s).sum() * (v + 1) + q.square().sum()).backward()
    assert torch.allclose(got["grad"], fp.flat_grad, rtol=1e-5, atol=1e-6)
    assert got["g"].tolist() == [3.0] * 5 and got["vis"].tolist() == [3] * 5 and got["mx"].tolist() == [1.0] * 5
    assert got["seed"] == got1["seed"] == 1234 and torch.equal(got["r"], got1["r"])


def _forced_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", FG_DP_FORCE_COLLECTIVES="1")
    from freegaussian_amd import viewdp

    assert viewdp._collective_world() == (1, False)  # no process group: nothing to force
    dist.init_process_group("gloo", rank=0, world_size=1)
    assert viewdp._collective_world() == (1, True)
    sc = synthetic_scene(50, 64, 64)
    fp = FlatGaussianParams.from_scene(sc, "cpu")
    m, q, s, o, c = fp.raster_inputs()
    ((m * 2).sum() + (c * 3).sum() + o.sum()).backward()
    before = fp.flat_grad.clone()
    fp.all_reduce_grads(average=True)  # one rank: the sum is the identity, and nothing is divided
    g, vis, mx = torch.full((5,), 2.0), torch.full((5,), 3), torch.full((5,), 4.0)
    all_reduce_densify_stats(g, vis, mx)
    seed = shared_seed(77)
    os.environ["FG_DP_FORCE_COLLECTIVES"] = "0"
    assert viewdp._collective_world() == (1, False)
    torch.save({"same": torch.equal(before, fp.flat_grad), "g": g, "vis": vis, "mx": mx, "seed": seed}, out)
    dist.destroy_process_group()


def test_forced_collectives_on_one_rank_are_the_identity(tmp_path):
    """FG_DP_FORCE_COLLECTIVES=1: the world > 1 code paths on a 1-rank group (the GPU form: scripts/rccl_one_rank.py)."""
    out = str(tmp_path / "f.pt")
    mp.spawn(_forced_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    assert got["same"] and got["g"].tolist() == [2.0] * 5 and got["vis"].tolist() == [3] * 5 and got["mx"].tolist() == [4.0] * 5
    assert got["seed"] == 77
