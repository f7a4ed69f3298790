"""CPU: host glue (utils, MLPs, flow conventions) against golden vectors produced by running the
reference's own Python (tests/golden/make_golden.py; data only)."""
import os
import sys

import numpy as np
import pytest
import torch

from freegaussian_amd import deform as D
from freegaussian_amd import flow as FL
from freegaussian_amd import utils as U
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig
from freegaussian_amd.rasterization import num_sh_bases, quat_to_rotmat
from oracle import raster_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_golden import fill_params  # noqa: E402  (pure helper; importing does not touch /root/reference)


def _load(name):
    z = np.load(os.path.join(GOLD, name))
    return lambda k: torch.from_numpy(z[k])


def test_utils_match_reference_goldens():
    t = _load("g_utils.npz")
    assert torch.equal(U.get_viewmat(t("viewmat_c2w")), t("viewmat_out"))
    assert torch.allclose(U.exp_se3(t("se3_S"), t("se3_theta")), t("se3_out"), atol=1e-7)
    e3, n3 = U.get_embedder(10, 3)
    e1, n1 = U.get_embedder(6, 1)
    e1b, n1b = U.get_embedder(10, 1)
    assert [n3, n1, n1b] == t("emb_dims").tolist() == [63, 13, 21]
    assert torch.equal(e3(t("emb_x3")), t("emb_x3_out")) and torch.equal(e1(t("emb_t1")), t("emb_t1_out"))
    assert torch.equal(e1b(t("emb_t1")), t("emb_t1_10_out"))
    assert torch.allclose(U.RGB2SH(t("rgb")), t("rgb2sh")) and torch.allclose(U.SH2RGB(t("rgb")), t("sh2rgb"))
    assert torch.allclose(U.resize_image(t("img"), 2), t("img_d2"), atol=1e-7)
    assert torch.allclose(U.resize_image(t("img"), 4), t("img_d4"), atol=1e-7)
    assert torch.equal(U.to_homogenous(t("hom_v")), t("hom_to"))
    assert torch.allclose(U.from_homogenous(torch.cat([t("hom_v"), torch.full((3, 1), 2.0)], -1)), t("hom_from"))


def test_bilinear_interp_reference_quirk_is_recorded_not_default():
    t = _load("g_utils.npz")
    img, x, y = t("bil_img"), t("bil_x"), t("bil_y")
    assert torch.allclose(U.bilinear_interp(img, x, y, reference_quirk=True), t("bil_out"))
    # integer coordinates: the reference returns 0 (floor == ceil), true bilinear returns the texel
    assert float(t("bil_out")[0, 3].abs().max()) == 0.0
    assert torch.equal(U.bilinear_interp(img, x, y)[0, 3], img[0, 1, 2])


@pytest.mark.parametrize("tag,kw", [("deform", {}), ("deform_blender", {"is_blender": True})])
def test_deform_mlp_matches_reference(tag, kw):
    t = _load("g_mlp.npz")
    m = D.FreeGaussianDeformableModel(**kw)
    fill_params(m)
    assert len(m.state_dict()) == int(t(tag + ".keys")[0])
    for ti, tt in enumerate((0.0, 0.5, 1.0)):
        dx, rot, sc = m(t("x"), torch.full((16, 1), tt))
        assert torch.allclose(dx, t(f"{tag}.t{ti}.d_xyz"), atol=1e-6)
        assert torch.allclose(rot, t(f"{tag}.t{ti}.rot"), atol=1e-6)
        assert torch.allclose(sc, t(f"{tag}.t{ti}.scale"), atol=1e-6)
    assert dx.shape == (16, 4, 4)


def test_control_mlp_matches_reference_and_state_dict_names():
    t = _load("g_mlp.npz")
    m = D.FreeGaussianControllableModel()
    fill_params(m)
    dx, rot, sc = m(t("x"), t("control.value"))
    assert torch.allclose(dx, t("control.d_xyz"), atol=1e-6) and torch.allclose(rot, t("control.rot"), atol=1e-6)
    assert torch.allclose(sc, t("control.scale"), atol=1e-6)
    keys = set(m.state_dict())
    assert {"linear.0.weight", "linear.7.bias", "d_xyz.weight", "d_scale.bias", "d_rot.weight"} <= keys
    dkeys = set(D.FreeGaussianDeformableModel(is_blender=True).state_dict())
    assert {"timenet.0.weight", "timenet.2.bias", "branch_w.weight", "branch_v.bias", "gaussian_rotation.weight",
            "gaussian_scaling.bias", "linear.5.weight"} <= dkeys  # fmt: skip


@pytest.mark.parametrize("tag", ["trans", "rot"])
def test_camera_flow_matches_reference_epipolar_flow(tag):
    t = _load("g_flow.npz")
    fx, fy, cx, cy = t("K").tolist()
    v, w = FL.relative_camera_motion(t("c2w0"), t(tag + ".c2w1"))
    sf = O.camera_flow(t("Z")[..., 0], fx, fy, cx, cy, v, w)
    assert torch.allclose(sf, t(tag + ".sceneflow"), atol=1e-6)
    inter = sf + t("opticalflow").double()
    inter[torch.isinf(t("Z")[..., 0])] = 0.0
    assert torch.allclose(inter, t(tag + ".interflow").double(), atol=1e-6)
    assert float(sf[1, 2].abs().max()) == 0.0  # infinite depth -> 0


@pytest.mark.parametrize("tag", ["trans", "rot", "small_trans", "small_rot"])
def test_reprojection_flow_restatement_matches_reference_epipolar_flow_bp(tag):
    """F-spec' (preprocess/epipolar_flow_bp.py:258-298): the oracle restatement against the output
    of the reference's own function (tests/golden/g_flow_bp.npz)."""
    t = _load("g_flow_bp.npz")
    fx, fy, cx, cy = t("K").tolist()
    K = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
    out = O.camera_flow_reprojection(t("Z"), t("Z1"), t(tag + ".c2w0"), t(tag + ".c2w1"), K, t("opticalflow"))
    assert torch.allclose(out["sceneflow"], t(tag + ".sceneflow"), atol=5e-6)
    assert torch.allclose(out["interflow"], t(tag + ".interflow"), atol=5e-6)
    assert float(out["sceneflow"][1, 2].abs().max()) == 0.0 and float(out["interflow"][1, 2].abs().max()) == 0.0
    # the host-side matrix of the product is the one the restatement applies
    M = FL.reprojection_motion(t(tag + ".c2w0"), t(tag + ".c2w1"))
    if tag == "trans":  # a translated camera: identity rotation, the WORLD-frame offset as translation
        assert torch.allclose(M[:, :3], torch.eye(3, dtype=torch.float64), atol=1e-6)
        assert torch.allclose(M[:, 3], (t("trans.c2w1") - t("trans.c2w0"))[:, 3].double(), atol=1e-6)


@pytest.mark.parametrize("tag", ["small_trans", "small_rot", "both"])
def test_reprojection_flow_agrees_with_the_AB_jacobian_to_first_order(tag):
    """The second, independent check of the A/B Jacobian's sign and layout (SURVEY.md section 8a,
    F-spec'): for a small camera motion the reprojection's `uv - xy` equals A v / Z + B w with
    v = M[:, 3] and w = rotation vector of M[:, :3] (M = the 3x4 matrix the reprojection applies),
    up to second order -- when the depth of frame 1 is the depth of the moved point."""
    from scipy.spatial.transform import Rotation as R

    t = _load("g_flow_bp.npz")
    fx, fy, cx, cy = t("K").tolist()
    K = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float64)
    if tag == "both":
        c0 = t("small_trans.c2w0")
        c1 = t("small_rot.c2w1").clone()
        c1[:, 3] = t("small_trans.c2w1")[:, 3]
    else:
        c0, c1 = t(tag + ".c2w0"), t(tag + ".c2w1")
    Z = t("Z").double()
    Z[torch.isinf(Z)] = 2.0
    M = FL.reprojection_motion(c0, c1)
    v = M[:, 3]
    w = torch.from_numpy(R.from_matrix(M[:, :3].numpy()).as_rotvec())
    # consistent frame-1 depth: z of the moved point
    H, W = Z.shape[:2]
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    p = torch.stack([(xx - cx) / fx, (yy - cy) / fy, torch.ones_like(xx)], -1) * Z
    Z1 = (p @ M[:, :3].T + M[:, 3])[..., 2:3]
    raw = O.camera_flow_reprojection(Z, Z1, c0, c1, K)["raw"]
    ab = O.camera_flow(Z[..., 0], fx, fy, cx, cy, v, w)
    scale = float(ab.abs().max())
    assert scale > 1e-3
    assert float((raw - ab).abs().max()) < 0.02 * scale  # second-order remainder at |motion| ~ 3e-3
    assert float((raw + ab).abs().max()) > 1.5 * scale  # ... and the sign is not the other one


def test_query_3d_gaussian_flow_matches_the_references_dead_code():
    """F-dead (freegaussian_model.py:662-751; zero call sites upstream): both variants against the
    output of the reference's own methods, quirks included; and the quirk-free mode differs exactly
    where the quirks bite."""
    t = _load("g_flow_query.npz")
    args = (t("means2d"), t("Z0"), t("interflow"), t("c2w1"), t("K"))
    assert torch.equal(FL.query_3d_gaussian_flow(*args)["p1_3d2"], t("plain"))
    assert torch.equal(FL.query_3d_gaussian_flow(*args, grid_size=16, step=8)["p1_3d2"], t("grid_16_8"))
    assert torch.equal(FL.query_3d_gaussian_flow(*args, grid_size=8, step=2)["p1_3d2"], t("grid_8_2"))
    on = ((t("means2d") >= 0) & (t("means2d") < torch.tensor([40.0, 24.0]))).all(-1)[0]
    assert 0 < int(on.sum()) < 30 and float(t("plain")[0][~on].abs().max()) == 0.0  # off-screen rows stay 0
    fixed = FL.query_3d_gaussian_flow(*args, reference_quirk=False)["p1_3d2"]
    assert float(fixed[0, 3].abs().max()) > 0.0  # integer coordinates: a real sample instead of the quirk's 0 weights
    assert not torch.equal(fixed[0, 3], t("plain")[0, 3])


def test_flow_AB_sign_convention_is_the_codes():
    A, B = O.camera_flow_AB(torch.tensor([3.0]), torch.tensor([1.0]), 7.0, 9.0, 2.5, 1.5)
    assert A[0].tolist() == [[7.0, 0.0, -0.5], [0.0, 9.0, 0.5]]  # +fx, cx - x: -1x the project page
    assert B[0, 0, 1].item() == pytest.approx(7.0 + 0.25 / 7.0)


def test_small_matrix_products_equal_bmm_in_value_and_gradient():
    """utils.small_bmm / transform_points (broadcast multiply + sum: what exp_se3 and the means' transform use instead
    of N-batch torch.bmm, freegaussian_model.py:840-843) against bmm and the reference's to_homogenous / from_homogenous
    composition, in double precision, with gradients."""
    from freegaussian_amd.utils import from_homogenous, small_bmm, to_homogenous, transform_points

    g = torch.Generator().manual_seed(2)
    for shape in ((5, 3, 3, 3), (7, 3, 3, 1), (4, 4, 4, 4), (1, 2, 5, 3)):
        n, a, b, c = shape
        A = torch.randn(n, a, b, generator=g, dtype=torch.float64, requires_grad=True)
        B = torch.randn(n, b, c, generator=g, dtype=torch.float64, requires_grad=True)
        w = torch.randn(n, a, c, generator=g, dtype=torch.float64)
        out = small_bmm(A, B)
        ref = torch.bmm(A, B)
        assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-13, atol=1e-13)
        ga = torch.autograd.grad((out * w).sum(), (A, B))
        gb = torch.autograd.grad((ref * w).sum(), (A, B))
        assert all(torch.allclose(x, y, rtol=1e-13, atol=1e-13) for x, y in zip(ga, gb))
    T = torch.randn(9, 4, 4, generator=g, dtype=torch.float64, requires_grad=True)
    pts = torch.randn(9, 3, generator=g, dtype=torch.float64, requires_grad=True)
    out = transform_points(T, pts)
    ref = from_homogenous(torch.bmm(T, to_homogenous(pts).unsqueeze(-1)).squeeze(-1))
    assert torch.allclose(out, ref, rtol=1e-12, atol=1e-12)
    ga = torch.autograd.grad(out.square().sum(), (T, pts))
    gb = torch.autograd.grad(ref.square().sum(), (T, pts))
    assert all(torch.allclose(x, y, rtol=1e-11, atol=1e-11) for x, y in zip(ga, gb))


def test_gsplat_helper_replacements():
    assert [num_sh_bases(d) for d in range(4)] == [1, 4, 9, 16]
    q = torch.tensor([[2.0, 0, 0, 0], [0.5, 0.5, 0.5, 0.5]])
    R = quat_to_rotmat(q)
    assert torch.allclose(R[0], torch.eye(3))
    assert torch.allclose(R[1], torch.tensor([[0.0, 0, 1], [1, 0, 0], [0, 1, 0]]), atol=1e-6)
    assert torch.allclose(R, O.quat_to_rotmat(q), atol=1e-6)


def test_model_schedules_and_camera_rescale():
    m = FreeGaussianModel(FreeGaussianModelConfig(), num_points=10)
    m.train()
    assert [m._get_downscale_factor() for m.step in (0, 2999, 3000, 5999, 6000, 30000)] == [4, 4, 2, 2, 1, 1]
    m.eval()
    assert m._get_downscale_factor() == 1
    m.step = 2500
    colors, deg = m._colors_and_degree()
    assert colors.shape == (10, 16, 3) and deg == 2
    cam = Camera(torch.eye(4)[None, :3], 1000.0, 1000.0, 960.0, 540.0, 1920, 1080)
    cam.rescale_output_resolution(0.25)
    assert (cam.width, cam.height, cam.fx, cam.cx) == (480, 270, 250.0, 240.0)
    g = m.get_param_groups()
    assert set(g) == {"means", "scales", "quats", "features_dc", "features_rest", "opacities", "deform", "control"}
    m.config.background_color = "bogus"
    with pytest.raises(ValueError):
        m._get_background_color()


@pytest.mark.parametrize("W,H", [(1352, 1014), (1920, 1080), (1015, 677), (128, 128)])
def test_rescaled_camera_size_equals_ground_truth_size(W, H):
    """The render size at the scheduled resolution must be the size get_gt_img gives the target
    (resize_image: H // d, reference utils.py:248-261) also when H % d >= d / 2."""
    m = FreeGaussianModel(FreeGaussianModelConfig(), num_points=4)
    m.train()
    img = torch.rand(H, W, 3)
    for m.step in (0, 3000, 6000):
        d = m._get_downscale_factor()
        cam = Camera(torch.eye(4)[None, :3], 800.0, 800.0, W / 2, H / 2, W, H)
        cam.rescale_output_resolution(1 / d)
        gt = m.get_gt_img(img)
        assert (cam.height, cam.width) == tuple(gt.shape[:2]) == (H // d, W // d)
        cam.rescale_output_resolution(d)
        if W % d == 0 and H % d == 0:
            assert (cam.width, cam.height) == (W, H)


def test_crop_box_selection_and_empty_outputs():
    """Eval-only crop (reference :778-798) and get_empty_outputs (:641-646); the raster call
    itself needs a GPU (tests/test_gpu_parity.py::test_crop_box_render)."""
    from freegaussian_amd.model import OrientedBox

    m = FreeGaussianModel(FreeGaussianModelConfig(), seed_points=torch.tensor([[0.0, 0, 0], [0.4, 0, 0], [3.0, 0, 0]]))
    box = OrientedBox(R=torch.eye(3), T=torch.zeros(3), S=torch.tensor([1.0, 1.0, 1.0]))
    m.set_crop(box)
    m.train()
    assert m._crop_ids() is None  # training ignores the crop box
    m.eval()
    assert m._crop_ids().tolist() == [True, True, False]
    m._active_crop = m._crop_ids()
    assert m.means.shape == (2, 3) and m.features_rest.shape == (2, 15, 3) and m.num_points == 3
    m._active_crop = None
    m.set_crop(OrientedBox(R=torch.eye(3), T=torch.full((3,), 50.0), S=torch.ones(3)))
    cam = Camera(torch.eye(4)[None, :3], 100.0, 100.0, 8.0, 6.0, 16, 12)
    out = m.get_outputs(cam)  # nothing inside: no raster call, hence fine on CPU
    assert out["rgb"].shape == (12, 16, 3) and float(out["depth"].min()) == 10.0 and float(out["accumulation"].max()) == 0.0
    # a rotated box: 45 degrees about z, long axis along the world diagonal
    c = 2**-0.5
    rot = OrientedBox(R=torch.tensor([[c, -c, 0.0], [c, c, 0.0], [0.0, 0.0, 1.0]]), T=torch.zeros(3), S=torch.tensor([4.0, 0.2, 1.0]))
    pts = torch.tensor([[1.0, 1.0, 0.0], [1.0, -1.0, 0.0]])
    assert rot.within(pts).reshape(-1).tolist() == [True, False]


def test_method_specs_mirror_reference_tables():
    from freegaussian_amd.method_config import METHODS, STAGE1_OPTIMIZERS, STAGE2_OPTIMIZERS, nerfstudio_method_specs

    assert set(METHODS) == {"freegaussian", "freegaussian-control"}  # pyproject.toml:14-17
    assert STAGE1_OPTIMIZERS["means"].lr == pytest.approx(8e-4) and STAGE1_OPTIMIZERS["means"].lr_final == pytest.approx(8e-6)
    assert STAGE1_OPTIMIZERS["features_rest"].lr == pytest.approx(0.0025 / 20)
    assert STAGE1_OPTIMIZERS["control"].max_steps == 15000 and "deform" not in STAGE2_OPTIMIZERS
    assert set(STAGE2_OPTIMIZERS) == set(STAGE1_OPTIMIZERS) - {"deform"}
    with pytest.raises(ImportError):
        nerfstudio_method_specs()  # nerfstudio is not installed here


def test_harness_loss_ssim_and_schedule():
    from freegaussian_amd import harness as Hn

    g = torch.Generator().manual_seed(0)
    a = torch.rand(40, 56, 3, generator=g)
    assert float(Hn.ssim(a.permute(2, 0, 1)[None], a.permute(2, 0, 1)[None])) == pytest.approx(1.0, abs=1e-6)
    b = (a + 0.1 * torch.randn(40, 56, 3, generator=g)).clamp(0, 1)
    s = float(Hn.ssim(a.permute(2, 0, 1)[None], b.permute(2, 0, 1)[None]))
    assert 0.3 < s < 0.999
    assert float(Hn.main_loss(a, a)) == pytest.approx(0.0, abs=1e-6)
    assert float(Hn.main_loss(b, a)) == pytest.approx(0.8 * (a - b).abs().mean().item() + 0.2 * (1 - s), rel=1e-5)
    m = FreeGaussianModel(FreeGaussianModelConfig(), num_points=8)
    opts = Hn.build_optimizers(m)
    assert set(opts) == {"means", "features_dc", "features_rest", "opacities", "scales", "quats", "deform", "control"}
    Hn.apply_schedules(opts, 15000)
    assert opts["means"].param_groups[0]["lr"] == pytest.approx((8e-4 * 8e-6) ** 0.5, rel=1e-6)
    assert opts["opacities"].param_groups[0]["lr"] == 0.05


def test_loss_dict_mirrors_the_reference_expressions():
    """get_loss_dict / composite_with_background / get_metrics_dict (freegaussian_model.py:911-990): RGBA ground
    truth over the step's background, the mask blacking out both images, main loss from L1 and SSIM with
    ssim_lambda, the scale-ratio regulariser 0.1 mean(max(ratio, max_gauss_ratio) - max_gauss_ratio) on every
    10th step only -- spelled out again here from the reference's lines."""
    from freegaussian_amd.harness import ssim

    g = torch.Generator().manual_seed(5)
    cfg = FreeGaussianModelConfig(num_downscales=0, use_scale_regularization=True, max_gauss_ratio=3.0)
    model = FreeGaussianModel(cfg, seed_points=torch.randn(50, 3, generator=g))
    with torch.no_grad():
        model.gauss_params["scales"].copy_(torch.randn(50, 3, generator=g))
    model.train()
    H, W = 24, 32
    pred, bg = torch.rand(H, W, 3, generator=g), torch.rand(3, generator=g)
    rgba = torch.rand(H, W, 4, generator=g)
    mask = (torch.rand(H, W, 1, generator=g) > 0.3).float()
    outputs = {"rgb": pred, "background": bg}
    gt = rgba[..., 3:] * rgba[..., :3] + (1 - rgba[..., 3:]) * bg  # (:918-920)
    assert torch.allclose(model.composite_with_background(rgba, bg), gt)
    assert model.composite_with_background(pred, bg) is pred
    model.step = 20
    d = model.get_loss_dict(outputs, {"image": rgba, "mask": mask})
    g_m, p_m = gt * mask, pred * mask
    main = 0.8 * (g_m - p_m).abs().mean() + 0.2 * (1 - ssim(g_m.permute(2, 0, 1)[None], p_m.permute(2, 0, 1)[None]))
    assert torch.allclose(d["main_loss"], main, atol=1e-7)
    se = torch.exp(model.scales)
    reg = 0.1 * (torch.maximum(se.amax(-1) / se.amin(-1), torch.tensor(3.0)) - 3.0).mean()  # (:968-976)
    assert float(reg) > 0 and torch.allclose(d["scale_reg"], reg)
    model.step = 21  # not a 10th step
    assert float(model.get_loss_dict(outputs, {"image": rgba})["scale_reg"]) == 0.0
    m = model.get_metrics_dict(outputs, {"image": rgba})
    assert abs(float(m["psnr"]) - float(-10 * torch.log10(((pred - gt) ** 2).mean()))) < 1e-5 and m["gaussian_count"] == 50
    # uint8 ground truth (:906-907)
    u8 = (torch.rand(H, W, 3, generator=g) * 255).to(torch.uint8)
    assert torch.allclose(model.get_gt_img(u8), u8.float() / 255.0)


def test_model_load_state_dict_resizes_and_remaps_like_the_reference():
    """freegaussian_model.py:278-291: parameters re-allocated to the checkpoint's count, legacy
    names remapped, step forced to 30000."""
    from freegaussian_amd.model import FreeGaussianModel, FreeGaussianModelConfig

    torch.manual_seed(0)
    src = FreeGaussianModel(FreeGaussianModelConfig(), num_points=70)
    dst = FreeGaussianModel(FreeGaussianModelConfig(), num_points=5)
    dst.step = 12
    dst.load_state_dict(src.state_dict())
    assert dst.num_points == 70 and dst.step == 30000
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    legacy = {k.replace("gauss_params.", ""): v for k, v in src.state_dict().items()}
    dst2 = FreeGaussianModel(FreeGaussianModelConfig(), num_points=9)
    dst2.load_state_dict(legacy)
    assert dst2.num_points == 70 and torch.equal(dst2.means, src.means)


def test_list_capacity_is_one_size_per_shape():
    """ops.list_capacity_for: the speculative capacity of the intersection lists covers the heaviest of the
    recent views with 25% headroom and does not change when the views differ by a few percent (every
    list-sized buffer of a step is allocated from it: a new size per view would churn the allocator)."""
    from freegaussian_amd.ops import list_capacity_for

    base = 5_085_668
    ring = [int(base * f) for f in (1.0, 0.985, 1.012, 0.97, 1.03, 0.99, 1.02, 0.975)]
    caps = {list_capacity_for(ring[: i + 1][-16:]) for i in range(2, len(ring))} | {list_capacity_for(ring)}
    assert len(caps) <= 2  # (the first views may still grow it once)
    cap = list_capacity_for(ring)
    assert cap >= 1.25 * max(ring) and cap <= 1.25 * max(ring) * (1 + 1 / 16) + 8192
    granule = 1 << max((int(max(ring) * 1.25) + 4096).bit_length() - 5, 12)
    assert cap % granule == 0
    assert list_capacity_for([0]) >= 4096 and list_capacity_for([2**31]) == 2**31 - 1
    assert list_capacity_for([10, 1_000_000, 10]) == list_capacity_for([1_000_000])
