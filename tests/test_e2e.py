"""End-to-end training (scripts/train_e2e.py): the procedural target, the reference's initialisation, a short run through
the HIP path whose held-out PSNR must rise, and the TWIN -- the same tiny training run through the HIP path and through the
CPU oracle from identical seeds (reference loop: freegaussian_pipeline.py:53-66, model freegaussian_model.py:150-196,
:753-898, :944-983)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from freegaussian_amd.model import FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import load_trained_scene, room_scene  # noqa: E402
from freegaussian_amd.utils import knn_mean_distance  # noqa: E402


def test_room_scene_is_deterministic_and_made_of_surfaces():
    a, ma = room_scene(20_000, 320, 180)
    b, mb = room_scene(20_000, 320, 180)
    for k in ("means", "quats", "scales", "opacities", "colors", "viewmats", "Ks"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert ma == mb and len(ma["train"]) == 32 and len(ma["test"]) == 8 and not set(ma["train"]) & set(ma["test"])
    assert a.means.shape == (20_000, 3) and a.colors.shape == (20_000, 16, 3)
    assert torch.allclose(a.quats.norm(dim=1), torch.ones(20_000), atol=1e-5)
    # flat or elongated, never round: the thin axis of a surface Gaussian is a tenth of its extent, a rod is long
    ratio = a.scales.max(dim=1).values / a.scales.min(dim=1).values
    assert float(ratio.min()) > 3.0
    # cameras stand inside the room (the walls are at |x|, |z| = 4), some of them among the objects
    eyes = -(a.viewmats[:, :3, :3].transpose(1, 2) @ a.viewmats[:, :3, 3:]).squeeze(-1)
    assert float(eyes[:, [0, 2]].abs().max()) < 4.0 and int((eyes[:, [0, 2]].norm(dim=1) < 2.0).sum()) >= 6
    # the third column of every surface Gaussian's rotation is the surface normal (floor: +y)
    name, first, cnt = ma["parts"][0]
    assert name == "floor"
    w, x, y, z = a.quats[first : first + cnt].unbind(1)
    col2 = torch.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], 1)
    assert torch.allclose(col2, torch.tensor([0.0, 1.0, 0.0]).expand_as(col2), atol=1e-5)


def test_initial_scales_are_the_references_knn_distances():
    """freegaussian_model.py:158-162 + :293-311: NearestNeighbors(k + 1), the point itself dropped, mean of three."""
    torch.manual_seed(0)
    m = FreeGaussianModel(FreeGaussianModelConfig(num_random=500))
    x = m.gauss_params["means"].detach()
    assert float(x.abs().max()) <= 5.0  # (rand - 0.5) * random_scale
    d = torch.cdist(x.double(), x.double())
    want = d.topk(4, dim=1, largest=False).values[:, 1:].mean(dim=1, keepdim=True).float().log().repeat(1, 3)
    assert torch.allclose(m.gauss_params["scales"].detach(), want, atol=1e-5)
    assert torch.allclose(knn_mean_distance(x[:2]), (x[0] - x[1]).norm().reshape(1, 1).expand(2, 1))  # fewer than k + 1 points
    assert float(torch.sigmoid(m.gauss_params["opacities"]).mean()) == pytest.approx(0.1, abs=1e-6)
    m2 = FreeGaussianModel(FreeGaussianModelConfig(), num_points=30, init_scales=-4.0)
    assert float(m2.gauss_params["scales"].max()) == -4.0


def test_trained_scene_file_round_trip(tmp_path):
    sc, _ = room_scene(500, 64, 48, n_views=10)
    p = str(tmp_path / "t.npz")
    np.savez(p, means=sc.means.numpy(), quats=sc.quats.numpy(), scales=sc.scales.numpy(), opacities=sc.opacities.numpy(),
             colors=sc.colors.numpy().astype("float16"), viewmats=sc.viewmats[:8].numpy(), Ks=sc.Ks[:8].numpy(), width=64, height=48,
             sh_degree=3, step=7)  # fmt: skip
    back = load_trained_scene(p)
    assert back.width == 64 and back.height == 48 and back.viewmats.shape == (8, 4, 4) and back.colors.dtype == torch.float32
    assert torch.equal(back.means, sc.means) and torch.allclose(back.colors, sc.colors, atol=2e-3)


# ------------------------------------------------------------------------------------------------
# GPU


class _OracleTwin(FreeGaussianModel):
    """The model on the host with the CPU oracle where the HIP rasterizer is: same kwargs as `_rasterize_and_finish`."""

    def _render(self, means, d_rotation, d_scaling, viewmat, K, W, H):
        from oracle import raster_oracle as O

        colors, sh_degree = self._colors_and_degree()
        scales = torch.exp(self.scales) + d_scaling
        quats = self.quats / self.quats.norm(dim=-1, keepdim=True) + d_rotation
        render, alpha, info = O.rasterization(means, quats, scales, torch.sigmoid(self.opacities).squeeze(-1), colors, viewmat, K, W, H,
                                              tile_size=16, packed=False, near_plane=0.01, far_plane=1e10, render_mode=self._render_mode(),
                                              sh_degree=sh_degree, sparse_grad=False, absgrad=True, rasterize_mode=self.config.rasterize_mode)  # fmt: skip
        if self.training and info["means2d"].requires_grad:
            info["means2d"].retain_grad()
        self.xys, self.radii = info["means2d"], info["radii"][0]
        background = self._get_background_color()
        rgb = torch.clamp(render[..., :3] + (1 - alpha) * background, 0.0, 1.0)
        return {"rgb": rgb.squeeze(0), "depth": None, "accumulation": alpha.squeeze(0), "background": background}


@pytest.mark.gpu
def test_twin_training_run_hip_vs_cpu_oracle():
    """64x64, 2000 Gaussians from random_init, 300 steps on a compressed schedule (1/4 -> 1/2 -> full resolution at 100 / 200,
    SH degree up every 60 steps, the deformation net from step 200): the loss of every one of the first 50 steps within 1 % of
    the CPU oracle's run from the same seeds.  (Two fp32 implementations of a chaotic optimisation drift apart eventually; the
    later steps are reported, and must still tell the same story: within 10 % on average over the last 50.)"""
    import copy

    import train_e2e as E
    from freegaussian_amd import harness

    dev = torch.device("cuda", 0)
    W = H = 64
    scene, meta = room_scene(20_000, W, H, n_views=40, seed=3, focal=48.0)
    gts = E.render_ground_truth(scene, dev)
    cams = [E.camera_from_viewmat(scene.viewmats[v], scene.Ks[v], W, H, meta["times"][v]) for v in range(40)]
    cfg = FreeGaussianModelConfig(num_random=2000, background_color="white", resolution_schedule=100, sh_degree_interval=60, warm_up=200)
    torch.manual_seed(5)
    cpu = _OracleTwin(cfg).train()
    gpu = FreeGaussianModel(copy.deepcopy(cfg), num_points=2000)
    gpu.load_state_dict(cpu.state_dict())
    gpu.step = 0  # (load_state_dict jumps to 30000 like the reference's, :280)
    gpu = gpu.to(dev).train()
    o_cpu, o_gpu = harness.build_optimizers(cpu), harness.build_optimizers(gpu)
    g = torch.Generator().manual_seed(1)
    la, lb = [], []
    for step in range(1, 301):
        v = meta["train"][int(torch.randint(0, 32, (1,), generator=g))]
        a = harness.train_step(gpu, o_gpu, cams[v], gts[v], step)
        b = harness.train_step(cpu, o_cpu, cams[v], gts[v].cpu(), step)
        la.append(a["loss"]), lb.append(b["loss"])
    rel = [abs(x - y) / abs(y) for x, y in zip(la, lb)]
    print(f"twin: loss {lb[0]:.4f} -> {lb[-1]:.4f} (oracle), {la[0]:.4f} -> {la[-1]:.4f} (HIP); max rel diff steps 1-50: {max(rel[:50]):.2e}, "
          f"51-200: {max(rel[50:200]):.2e}, 201-300 (deform net on): {max(rel[200:]):.2e}")  # fmt: skip
    if os.environ.get("FG_TWIN_REPORT"):
        import json

        json.dump({"loss_hip": la, "loss_oracle": lb, "rel": rel}, open(os.environ["FG_TWIN_REPORT"], "w"))
    assert max(rel[:50]) < 0.01, max(rel[:50])
    assert lb[-1] < 0.8 * lb[0] and la[-1] < 0.8 * la[0]  # both runs learn
    assert sum(rel[-50:]) / 50 < 0.10


def _trained_frame_vs_oracle(model, scene, meta):
    """What the OPTIMISER produced -- anisotropic splats, some hundreds of pixels wide, some next to the lens -- through the HIP
    path and the CPU oracle: the reference's lists bit for bit, the walked lists an order-preserving subsequence of them, image
    and every gradient at the suite's bar.  One ring view and one view from among the objects."""
    from helpers import REL_TOL, close_except_knife_edge, rel_l2
    from oracle import raster_oracle as O

    from freegaussian_amd import rasterization

    gp = model.gauss_params
    with torch.no_grad():
        ins = [gp["means"], gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True), torch.exp(gp["scales"]),
               torch.sigmoid(gp["opacities"]).squeeze(-1), torch.cat([gp["features_dc"][:, None, :], gp["features_rest"]], 1)]  # fmt: skip
    ins = [t.detach().clone() for t in ins]
    W, H = scene.width, scene.height
    for v in (meta["kinds"][0][1], meta["kinds"][3][1] + 2):
        vm, K = scene.viewmats[v : v + 1], scene.Ks[v : v + 1]
        vr = torch.randn(1, H, W, 3, generator=torch.Generator().manual_seed(v))
        g = [t.clone().requires_grad_(True) for t in ins]
        r, a, info = rasterization(*g, vm.to(g[0].device), K.to(g[0].device), W, H, sh_degree=3, packed=False, absgrad=True)
        gh = torch.autograd.grad(r, g, vr.to(r.device), retain_graph=True)
        c = [t.cpu().clone().requires_grad_(True) for t in ins]
        r0, a0, info0 = O.rasterization(*c, vm, K, W, H, sh_degree=3, packed=False, absgrad=True)
        go = torch.autograd.grad(r0, c, vr, retain_graph=True)
        assert torch.equal(info["radii"].cpu(), info0["radii"]) and torch.equal(info["flatten_ids"].cpu(), info0["flatten_ids"]), v
        assert torch.equal(info["isect_offsets"].cpu(), info0["isect_offsets"])
        assert close_except_knife_edge(r, r0) and close_except_knife_edge(a, a0), v
        errs = [rel_l2(x.cpu(), y) for x, y in zip(gh, go)]
        worst = max(errs)
        print(f"trained frame, view {v}: V {int((info0['radii'] > 0).sum())} of {ins[0].shape[0]}, I {info0['flatten_ids'].numel()}, "
              f"largest radius {int(info0['radii'].max())} px, worst gradient rel L2 {worst:.1e}")
        if worst >= REL_TOL or os.environ.get("FG_TEST_FORCE_KNIFE_EDGE"):
            # The training run is not bit-reproducible (atomic sums): every run of the gate checks ANOTHER trained frame.  Two of
            # ~220 frames so far had a gradient 1.4e-4 from the oracle's where the others are at 2e-6 ... 1.3e-5
            # (scripts/e2e_parity_outlier.py): a discrete event, not noise.  The suite's rule for images -- a pixel whose alpha or
            # transmittance is within rounding of a threshold may take the other branch -- applied to the backward, which is
            # linear in the cotangent: with the cotangent zeroed on the pixels where the two IMAGES differ, few of them,
            # every gradient must meet the bar.
            import helpers

            scale = float(r0.detach().abs().max())
            d_img = (r.detach().cpu() - r0.detach()).abs().amax(-1)[0]
            flipped = d_img > 1e-5 * scale  # (ordinary pixels are ~1e-6 apart; a flipped 1/255 decision moves one by up to 4e-3)
            n_flipped = int(flipped.sum())
            keep = (~flipped).float()[None, :, :, None]
            gh2 = torch.autograd.grad(r, g, (vr * keep).to(r.device))
            go2 = torch.autograd.grad(r0, c, vr * keep)
            errs2 = [rel_l2(x.cpu(), y) for x, y in zip(gh2, go2)]
            helpers._record("knife_edge_gradient_pixels_masked", n_flipped)
            print(f"  beyond the bar ({worst:.1e}); with the cotangent zeroed on {n_flipped} knife-edge pixels: {max(errs2):.1e}")
            assert n_flipped <= 16 and max(errs2) < REL_TOL, (v, errs, n_flipped, errs2)


@pytest.mark.gpu
def test_short_end_to_end_run_learns_the_scene():
    """A bounded version of scripts/train_e2e.py inside the gate: 240x135 target, schedule compressed 10x (resolution 150 / 300,
    refinements every 20 steps from 50, opacity reset every 600), 700 steps from random_init.  Held-out PSNR rises and ends
    above 17 dB; densification ran; no stage-wise fallbacks beyond the shapes' first calls."""
    import train_e2e as E

    over = dict(resolution_schedule=150, sh_degree_interval=100, refine_every=20, refine_start=50, stop_screen_size_at=400,
                stop_split_at=1500)  # fmt: skip
    model, rep, _ = E.train(steps=700, n_target=30_000, width=240, height=135, seed=42, eval_at=(100, 300, 700), warm_up=10**9,
                            num_random=8000, log=lambda *a: None, config_overrides=over)  # fmt: skip
    ps = [e["heldout_psnr"] for e in rep["evals"]]
    print("short e2e: held-out PSNR", [round(p, 2) for p in ps], "N", [e["N"] for e in rep["evals"]])
    _trained_frame_vs_oracle(model, _[0], _[1])
    assert ps[0] < ps[1] < ps[2] and ps[2] > 17.0  # (19.3 ... 21.6 over 24 runs on one box: training is not bit-reproducible)
    assert rep["N_final"] > 8000
    assert rep["policy_counters"]["end"]["capacity_redos"] <= 6


@pytest.mark.gpu
def test_bench_trains_its_own_scene_when_the_checkpoint_file_is_absent():
    """`bench.py --layout trained:auto` without `data/trained_scene_r06.npz` (46 MB, not in the history -- the state of a fresh
    checkout): the child trains the scene in its own process before anything is timed and reports where the scene came from.
    A small run here (FG_BENCH_TRAIN); the default is scripts/train_e2e.py's 7000 steps at 1080p, ~10 s."""
    import json
    import subprocess

    env = dict(os.environ, FG_BENCH_TRAIN_HERE="1", FG_BENCH_TRAIN="400,20000,480,270,6000")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--layout", "trained:auto", "--steps", "12", "--warmup", "4",
                          "--settle-s", "0.2", "--no-cpu-baseline", "--no-graph", "--no-clustered"], capture_output=True, text=True, env=env, timeout=600)  # fmt: skip
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert res.returncode == 0 and lines, res.stderr[-2000:]
    d = json.loads(lines[-1])
    c = d["config"]
    assert c["layout"] == "trained:auto" and "trained in this process" in c["scene"]["source"] and c["scene"]["heldout_psnr_db"] > 12.0
    assert c["N"] == c["scene"]["N_final"] and 0 < c["V"] <= c["N"] and c["P"] == 480 * 270 and d["value"] > 0
    assert "axis_ratio_hist" in c["scene_statistics"] and d["path_events_in_timed_region"]["stagewise_raster_calls"] == 0
