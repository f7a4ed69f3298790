"""The optional bilateral-grid branch (freegaussian_amd/bilagrid.py; reference freegaussian_model.py:122-126, :227-233,
:617-618, :879-882, :935-937, :988-989).  The library the reference imports it from is not in the tree ("parity
unpinned" in the module's header): the checks are a scalar restatement of the trilinear slice and the properties the
algorithm defines."""
import math

import numpy as np
import pytest
import torch

from freegaussian_amd import bilagrid
from freegaussian_amd.bilagrid import BilateralGrid, color_correct, total_variation_loss


def _trilinear(grid, x, y, z):
    """grid [12, L, H, W]; x, y, z in [-1, 1], align_corners, border clamp: plain loops."""
    _, L, H, W = grid.shape

    def axis(v, n):
        f = min(max((v + 1) / 2 * (n - 1), 0.0), n - 1.0)
        i0 = min(int(math.floor(f)), n - 1)
        i1 = min(i0 + 1, n - 1)
        return i0, i1, f - i0

    x0, x1, fx = axis(x, W)
    y0, y1, fy = axis(y, H)
    z0, z1, fz = axis(z, L)
    out = np.zeros(12)
    for zi, wz in ((z0, 1 - fz), (z1, fz)):
        for yi, wy in ((y0, 1 - fy), (y1, fy)):
            for xi, wx in ((x0, 1 - fx), (x1, fx)):
                out += wz * wy * wx * grid[:, zi, yi, xi]
    return out


def test_identity_grids_leave_the_image_alone_and_have_no_variation():
    g = BilateralGrid(num=3, grid_X=4, grid_Y=5, grid_W=2)
    assert tuple(g.grids.shape) == (3, 12, 2, 5, 4)
    rgb = torch.rand(1, 9, 7, 3)
    out = bilagrid.apply_to_render(g, rgb, cam_idx=2, H=9, W=7)
    assert torch.allclose(out, rgb, atol=1e-6)
    assert float(g.tv_loss().detach()) == 0.0


def test_slice_is_trilinear_in_x_y_and_luma():
    torch.manual_seed(3)
    g = BilateralGrid(num=2, grid_X=5, grid_Y=4, grid_W=3)
    with torch.no_grad():
        g.grids.add_(0.3 * torch.randn_like(g.grids))
    H, W = 6, 8
    rgb = torch.rand(1, H, W, 3)
    out = bilagrid.apply_to_render(g, rgb, cam_idx=1, H=H, W=W)[0].detach().numpy()
    grid = g.grids[1].detach().numpy().astype(np.float64)
    for py in range(H):
        for px in range(W):
            c = rgb[0, py, px].numpy().astype(np.float64)
            luma = float(c @ np.array([0.299, 0.587, 0.114]))
            A = _trilinear(grid, px / (W - 1) * 2 - 1, py / (H - 1) * 2 - 1, luma * 2 - 1).reshape(3, 4)
            np.testing.assert_allclose(out[py, px], A[:, :3] @ c + A[:, 3], atol=2e-5)


def test_slice_accepts_flat_pixel_lists_and_returns_the_matrices():
    g = BilateralGrid(num=1, grid_X=3, grid_Y=3, grid_W=2)
    xy, rgb = torch.rand(17, 2), torch.rand(17, 3)
    out = bilagrid.slice(g, xy, rgb, torch.tensor([0]))
    assert tuple(out["rgb"].shape) == (17, 3) and tuple(out["rgb_affine_mats"].shape) == (17, 3, 4)
    assert torch.allclose(out["rgb"], rgb, atol=1e-6)


def test_total_variation_counts_squared_steps_along_the_three_axes():
    x = torch.zeros(2, 1, 2, 2, 3)
    x[0, 0, 1] = 1.0  # a step of 1 along L in every (h, w) cell of grid 0
    # axis L: 6 cells x 1^2 / 6 values per batch item = 1; nothing along H or W; averaged over the 2 grids
    assert float(total_variation_loss(x)) == pytest.approx(0.5)
    y = torch.zeros(1, 1, 1, 1, 4)
    y[..., 2:] = 2.0  # one step of 2 along W: 4 / 3 differences
    assert float(total_variation_loss(y)) == pytest.approx(4.0 / 3.0)


def test_gradients_reach_only_the_grid_of_the_camera():
    g = BilateralGrid(num=3, grid_X=4, grid_Y=4, grid_W=2)
    rgb = torch.rand(1, 5, 5, 3, requires_grad=True)
    bilagrid.apply_to_render(g, rgb, cam_idx=1, H=5, W=5).square().sum().backward()
    per_grid = g.grids.grad.abs().sum(dim=(1, 2, 3, 4))
    assert per_grid[1] > 0 and per_grid[0] == 0 and per_grid[2] == 0 and rgb.grad.abs().sum() > 0


def test_color_correct_undoes_an_affine_colour_change():
    torch.manual_seed(0)
    ref = torch.rand(24, 20, 3) * 0.8 + 0.1
    M = torch.tensor([[0.9, 0.05, 0.0], [0.02, 0.85, 0.03], [0.0, 0.04, 0.8]])
    img = ref @ M.T + torch.tensor([0.03, 0.01, 0.05])
    cc = color_correct(img, ref)
    assert float((img - ref).abs().mean()) > 0.03
    assert float((cc - ref).abs().max()) < 2e-3
    with pytest.raises(ValueError):
        color_correct(img, ref[..., :2])


def test_model_switch_adds_the_group_the_loss_term_and_the_training_only_slice():
    from freegaussian_amd.harness import build_optimizers
    from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig

    cfg = FreeGaussianModelConfig(use_bilateral_grid=True, grid_shape=(4, 4, 2), color_corrected_metrics=True)
    with pytest.raises(ValueError):
        FreeGaussianModel(cfg, num_points=64, init_scales=-4.0)
    m = FreeGaussianModel(cfg, num_points=64, init_scales=-4.0, num_train_data=5)
    assert tuple(m.bil_grids.grids.shape) == (5, 12, 2, 4, 4)
    groups = m.get_param_groups()
    assert groups["bilateral_grid"] == [m.bil_grids.grids]
    assert "bilateral_grid" in build_optimizers(m)
    assert "bilateral_grid" not in build_optimizers(FreeGaussianModel(num_points=8, init_scales=-4.0))
    with torch.no_grad():
        m.bil_grids.grids[3, 3] += 0.25  # grid 3: +0.25 on red
    cam = Camera(torch.eye(4)[None, :3], 10.0, 10.0, 4.0, 4.0, 8, 8, times=torch.zeros(1, 1), metadata={"cam_idx": 3})
    rgb = torch.rand(8, 8, 3) * 0.5
    m.train()
    out = m._bilateral({"rgb": rgb.clone()}, cam)
    assert torch.allclose(out["rgb"][..., 0], rgb[..., 0] + 0.25, atol=1e-6) and torch.allclose(out["rgb"][..., 1:], rgb[..., 1:], atol=1e-6)
    cam.metadata.pop("cam_idx")
    assert torch.equal(m._bilateral({"rgb": rgb.clone()}, cam)["rgb"], rgb)  # (:881: no index, no slice)
    cam.metadata["cam_idx"] = 3
    m.eval()
    assert torch.equal(m._bilateral({"rgb": rgb.clone()}, cam)["rgb"], rgb)  # (:880: training only)
    # the loss dict: 10 x TV while training; the colour-corrected PSNR beside the plain one
    m.train()
    m.step = 30000
    rgb = torch.rand(16, 16, 3) * 0.8 + 0.1
    outputs = {"rgb": rgb, "background": torch.zeros(3)}
    batch = {"image": (rgb * 0.9 + 0.02)}
    ld = m.get_loss_dict(outputs, batch)
    assert float(ld["tv_loss"].detach()) == pytest.approx(10 * float(total_variation_loss(m.bil_grids.grids.detach())))
    md = m.get_metrics_dict(outputs, batch)
    assert float(md["cc_psnr"]) > float(md["psnr"]) + 10
    m.eval()
    assert "tv_loss" not in m.get_loss_dict(outputs, batch)


@pytest.mark.gpu
def test_training_with_bilateral_grids_learns_a_per_image_colour_cast():
    """Two training images of one tiny scene, the second with a colour cast the Gaussians cannot explain without breaking the
    first: with the switch on the cast goes into image 2's grid (its loss falls well below the run without grids) and an
    eval render, which never goes through a grid, stays closest to the uncast image."""
    from freegaussian_amd.harness import build_optimizers, train_step
    from freegaussian_amd.model import FreeGaussianModel, FreeGaussianModelConfig
    from freegaussian_amd.scenes import room_scene
    from scripts.train_e2e import camera_from_viewmat, render_ground_truth

    dev = torch.device("cuda", 0)
    W, H = 160, 96
    scene, meta = room_scene(6000, W, H, n_views=40, seed=2)
    scene.viewmats, scene.Ks = scene.viewmats[:1], scene.Ks[:1]
    rgba = render_ground_truth(scene, dev)[0].float() / 255.0
    gt = rgba[..., :3] * rgba[..., 3:]  # over black
    cast = (gt * torch.tensor([0.7, 1.0, 0.8], device=dev) + torch.tensor([0.15, 0.0, 0.1], device=dev)).clamp(0, 1)
    images = [gt, cast]

    def run(use_grid):
        torch.manual_seed(0)
        cfg = FreeGaussianModelConfig(use_bilateral_grid=use_grid, grid_shape=(8, 8, 4), background_color="black", num_downscales=0,
                                      warm_up=10**9, refine_every=10**9, num_random=4000, random_scale=3.0)  # fmt: skip
        m = FreeGaussianModel(cfg, num_train_data=2 if use_grid else None).to(dev).train()
        opts = build_optimizers(m)
        last = [None, None]
        for step in range(1, 241):
            i = step % 2
            cam = camera_from_viewmat(scene.viewmats[0].clone(), scene.Ks[0], W, H, 0.0)
            cam.metadata["cam_idx"] = i
            res = train_step(m, opts, cam, images[i], step)
            last[i] = res["loss"]
        m.eval()
        with torch.no_grad():
            ev = m.get_outputs(camera_from_viewmat(scene.viewmats[0].clone(), scene.Ks[0], W, H, 0.0))["rgb"]
        return m, last, ev

    m_on, last_on, ev_on = run(True)
    m_off, last_off, ev_off = run(False)
    print("bilateral grids: last losses with", [round(x, 4) for x in last_on], "without", [round(x, 4) for x in last_off])
    assert last_on[1] < 0.8 * last_off[1] and last_on[0] < 0.8 * last_off[0], (last_on, last_off)
    g = m_on.bil_grids.grids.detach()
    eye = torch.tensor([1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0], device=dev).view(1, 12, 1, 1, 1)
    assert float((g - eye).abs().max()) > 0.02  # the grids moved
    assert float((g[0] - g[1]).abs().max()) > 0.02  # ... apart: the cast is per image
    assert ev_on.shape == gt.shape and bool(torch.isfinite(ev_on).all())
