"""Stage-1 -> stage-2 hand-off: attribute-mask back-projection
(HIP vs the restated reference expressions) and the on-disk formats."""
import os

import numpy as np
import pytest
import torch

from freegaussian_amd import io as fio
from oracle import backproject_oracle as BO


def test_gaussian_mask_and_interflow_files_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    mask = torch.rand(500, 3, generator=g) > 0.7
    p = fio.save_gaussian_mask(str(tmp_path), mask)
    assert os.path.basename(p) == "gaussian_mask_NxM.npy"
    raw = np.load(p)
    assert raw.dtype == np.bool_ and raw.shape == (500, 3)  # what freegaussian_pipeline.py:47 feeds torch.from_numpy
    assert torch.equal(fio.load_gaussian_mask(str(tmp_path)), mask)
    assert os.path.basename(fio.save_gaussian_mask(str(tmp_path), mask, crop=True)) == "gaussian_mask_NxM_crop.npy"
    flow = torch.randn(12, 20, 2, generator=g)
    fp = fio.save_interflow(str(tmp_path), "./images/frame_00007.png", 5, flow)
    assert fp.endswith(os.path.join("interflow_n5", "frame_00007.png.npy"))  # freegaussian_dataparser.py:1165
    assert torch.equal(fio.load_interflow(str(tmp_path), "./images/frame_00007.png", 5), flow)


def test_checkpoint_layout_and_stage2_load(tmp_path):
    from freegaussian_amd.model import Camera, FreeGaussianControlModel, FreeGaussianModel, FreeGaussianModelConfig

    torch.manual_seed(0)
    stage1 = FreeGaussianModel(FreeGaussianModelConfig(), num_points=300)
    opts = {"means": torch.optim.Adam([stage1.gauss_params["means"]], lr=1e-3)}
    path = fio.save_checkpoint(str(tmp_path), 29999, stage1, opts)
    assert os.path.basename(path) == "step-000029999.ckpt"
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    assert set(loaded) >= {"step", "pipeline", "optimizers", "schedulers"} and loaded["step"] == 29999
    assert all(k.startswith("_model.") for k in loaded["pipeline"])
    cam = Camera(torch.eye(4)[None, :3], 100.0, 100.0, 32.0, 32.0, 64, 64, times=torch.zeros(1, 1))
    stage2 = FreeGaussianControlModel(torch.zeros(50, 2, dtype=torch.bool), cam, config=FreeGaussianModelConfig(),
                                      num_points=50)  # fmt: skip
    assert fio.load_deformable_checkpoint(stage2, path) == 29999
    assert stage2.num_points == 300
    for k, v in stage1.state_dict().items():
        assert torch.equal(stage2.state_dict()[k], v), k
    # DDP-wrapped writer: "module." in front of everything and inside "_model."
    ddp = dict(loaded)
    ddp["pipeline"] = {"module._model.module." + k[len("_model."):]: v for k, v in loaded["pipeline"].items()}
    p2 = os.path.join(str(tmp_path), "ddp.ckpt")
    torch.save(ddp, p2)
    st = fio.model_state_from_checkpoint(p2)
    assert set(st) == set(stage1.state_dict())


def _frame_labels(H, W, M, seed):
    g = torch.Generator().manual_seed(seed)
    atrb = torch.zeros(H, W, M + 1, dtype=torch.bool)
    for j in range(M):  # blobs of labels, overlapping
        cy, cx = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, W, (1,), generator=g))
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        atrb[..., j] = ((yy - cy) ** 2 + (xx - cx) ** 2) < (min(H, W) // 3) ** 2
    atrb[..., M] = ~atrb[..., :M].any(-1)
    valids = torch.ones(M + 1, dtype=torch.bool)
    valids[1] = False  # an attribute that is not annotated in this frame
    return atrb, valids


@pytest.mark.gpu
def test_mask_backprojection_matches_reference_expressions():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (no CPU fallback exists)")
    from freegaussian_amd import rasterization
    from freegaussian_amd.masks import backproject_frame, build_gaussian_masks
    from freegaussian_amd.scenes import synthetic_scene

    sc = synthetic_scene(20000, 240, 136, n_views=2, seed=13)
    sc.means[:200] *= 3.0  # some centres outside the image / behind other surfaces
    dev = "cuda"
    t = [x.to(dev) for x in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
    M = 4
    ref = torch.zeros(20000, M, dtype=torch.bool)
    frames = []
    for v in range(2):
        atrb, valids = _frame_labels(sc.height, sc.width, M, seed=v)
        vm, K = sc.viewmats[v : v + 1], sc.Ks[v : v + 1]
        frames.append((vm, K, sc.width, sc.height, atrb, valids))
        with torch.no_grad():
            render, _, info = rasterization(*t, vm.to(dev), K.to(dev), sc.width, sc.height, packed=True,
                                            render_mode="ED", sh_degree=3)  # fmt: skip
        BO.backproject_frame(ref, info["means2d"].cpu(), info["depths"].cpu(), info["gaussian_ids"].cpu(),
                             render.squeeze().cpu(), atrb, valids)  # fmt: skip
    out = build_gaussian_masks(*t, 3, frames)
    assert out.dtype == torch.bool and out.shape == (20000, M)
    assert 200 < int(ref.sum()) < 20000 * M and not bool(ref[:, 1].any() and False)
    assert torch.equal(out.cpu(), ref)
    # single-frame call on explicit arrays, incl. a centre in (-1, 0) that truncates onto pixel 0
    gm = torch.zeros(3, M, dtype=torch.bool, device=dev)
    m2 = torch.tensor([[-0.5, -0.25], [5.2, 7.9], [300.0, 2.0]], device=dev)
    depth_map = torch.full((sc.height, sc.width), 2.0, device=dev)
    atrb = torch.ones(sc.height, sc.width, M + 1, dtype=torch.bool)
    backproject_frame(gm, m2, torch.tensor([2.1, 4.5, 2.0], device=dev), torch.tensor([3, 3, 3], device=dev),
                      depth_map, atrb, torch.ones(M + 1, dtype=torch.bool))  # fmt: skip
    # row 0: inside after truncation, |delta| small -> labelled; row 1: delta = -2.5 < -0.2 -> rejected; row 2: outside
    assert gm.cpu().tolist() == [[True] * M, [False] * M, [False] * M]


@pytest.mark.gpu
def test_stage1_to_stage2_end_to_end(tmp_path):
    """The whole hand-off on a synthetic scene: stage-1 training steps with the
    deform net -> step-%09d.ckpt -> key-frame mask back-projection -> gaussian_mask_NxM.npy ->
    FreeGaussianControlModel initialised from both files -> stage-2 training steps that only move
    the control net."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no GPU is visible (no CPU fallback exists)")
    import copy
    import sys

    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _model_and_camera

    from freegaussian_amd import harness as Hn
    from freegaussian_amd.masks import build_gaussian_masks
    from freegaussian_amd.method_config import STAGE2_OPTIMIZERS
    from freegaussian_amd.model import FreeGaussianControlModel
    from freegaussian_amd.utils import get_viewmat

    dev = "cuda"
    model, _, cam = _model_and_camera(n=3000, W=128, H=96, step=3000, training=True)  # after warm-up: deform active
    target = copy.deepcopy(model).eval()
    with torch.no_grad():
        target.gauss_params["features_dc"].add_(0.2)
        gt = target.get_outputs(copy.deepcopy(cam))["rgb"].clamp(0, 1)
    opts = Hn.build_optimizers(model)
    h1 = [Hn.train_step(model, opts, copy.deepcopy(cam), gt, 3000 + i) for i in range(8)]
    assert h1[-1]["loss"] < h1[0]["loss"]
    ckpt = fio.save_checkpoint(str(tmp_path / "ckpt"), 3007, model, opts)

    # key frame: label the left and right halves of the image as two attributes (+ background)
    H, W = cam.height, cam.width
    atrb = torch.zeros(H, W, 3, dtype=torch.bool)
    atrb[:, : W // 2, 0] = True
    atrb[:, W // 2 :, 1] = True
    valids = torch.ones(3, dtype=torch.bool)
    gp = {k: v.detach() for k, v in model.gauss_params.items()}
    colors = torch.cat([gp["features_dc"][:, None], gp["features_rest"]], 1)
    frame = (get_viewmat(cam.camera_to_worlds), cam.get_intrinsics_matrices(), W, H, atrb, valids)
    masks = build_gaussian_masks(gp["means"], gp["quats"] / gp["quats"].norm(dim=-1, keepdim=True), gp["scales"].exp(),
                                 torch.sigmoid(gp["opacities"]).squeeze(-1), colors, 3, [frame])  # fmt: skip
    assert masks.shape == (3000, 2) and 50 < int(masks.any(-1).sum()) < 3000
    assert not bool((masks[:, 0] & masks[:, 1]).any())  # a centre lies in one half only
    fio.save_gaussian_mask(str(tmp_path), masks)

    # stage 2 from the two files
    init_cam = type(cam)(cam.camera_to_worlds, cam.fx, cam.fy, cam.cx, cam.cy, W, H, times=torch.tensor([[0.0]]))
    stage2 = FreeGaussianControlModel(fio.load_gaussian_mask(str(tmp_path)), init_cam, config=copy.deepcopy(model.config),
                                      num_points=10)  # fmt: skip
    assert fio.load_deformable_checkpoint(stage2, ckpt) == 3007 and stage2.num_points == 3000
    stage2 = stage2.to(dev).train()
    for k in model.gauss_params:
        assert torch.equal(stage2.gauss_params[k], model.gauss_params[k])
    opts2 = Hn.build_optimizers(stage2, STAGE2_OPTIMIZERS)
    assert "deform" not in opts2 and "control" in opts2
    before = [p.detach().clone() for p in stage2.deform.parameters()]
    ctrl0 = torch.cat([p.detach().flatten().clone() for p in stage2.control.parameters()])
    cam2 = copy.deepcopy(cam)
    cam2.metadata["cameras0"] = init_cam
    h2 = [Hn.train_step(stage2, opts2, copy.deepcopy(cam2), gt, 30000 + i, table=STAGE2_OPTIMIZERS) for i in range(6)]
    assert all(torch.isfinite(torch.tensor(h["loss"])) for h in h2)
    assert all(torch.equal(a, b) for a, b in zip(before, stage2.deform.parameters()))  # frozen in stage 2
    ctrl1 = torch.cat([p.detach().flatten() for p in stage2.control.parameters()])
    assert float((ctrl1 - ctrl0).abs().max()) > 0  # the control net is what trains
