#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's own Python.

Run in the build container only (needs /root/reference; never at test time):

    python tests/golden/make_golden.py

What is captured -- data only, no reference source text is stored:
  g_utils.npz   get_viewmat, exp_se3, Embedder (get_embedder), RGB2SH/SH2RGB, resize_image,
                bilinear_interp, to/from_homogenous        (freegaussian/utils.py)
  g_mlp.npz     FreeGaussianDeformableModel / FreeGaussianControllableModel forward outputs for
                a seeded state_dict                         (freegaussian_model.py:1054-1145)
  g_flow.npz    diff_2d_epipolar_flow sceneflow / interflow for a translation pair and a
                rotation pair                               (preprocess/epipolar_flow.py:212-321)
  g_flow_bp.npz the exact-reprojection variant (F-spec') on the same pairs + small-motion pairs
                                                            (preprocess/epipolar_flow_bp.py:229-298)
  g_flow_query.npz query_3d_gaussian_flow / query_3d_gaussian_flow_grid (dead code upstream)
                                                            (freegaussian_model.py:662-751)
  g_densify.npz refinement_after / split_gaussians / dup_gaussians / cull_gaussians / the
                optimizer surgery and after_train_iter, executed as methods of a stub ``self``
                                                            (freegaussian_model.py:313-392, :404-571)

The reference modules import nerfstudio / mmflow, which are not installed; utils.py is loaded
with a stub for its single non-torch import and the two other pieces are executed from their
AST slices with stubs for the pose helpers they call."""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_utils():
    ns = types.ModuleType("nerfstudio")
    nsu = types.ModuleType("nerfstudio.utils")
    misc = types.ModuleType("nerfstudio.utils.misc")
    misc.torch_compile = lambda *a, **k: (lambda f: f)
    sys.modules.update({"nerfstudio": ns, "nerfstudio.utils": nsu, "nerfstudio.utils.misc": misc})
    spec = importlib.util.spec_from_file_location("ref_utils", os.path.join(REF, "freegaussian", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def slice_defs(path, names):
    """Source of the named top-level defs/classes of a reference file (executed, never saved)."""
    src = open(path).read()
    tree = ast.parse(src)
    lines = src.splitlines()
    out = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out.append("\n".join(lines[node.lineno - 1 : node.end_lineno]))
    assert len(out) == len(names), (names, len(out))
    return "\n\n".join(out)


def gen_utils(U):
    g = torch.Generator().manual_seed(0)
    d = {}
    # G1 get_viewmat
    R = torch.linalg.qr(torch.randn(4, 3, 3, generator=g)).Q
    c2w = torch.cat([R, torch.randn(4, 3, 1, generator=g)], -1)
    d["viewmat_c2w"], d["viewmat_out"] = c2w, U.get_viewmat(c2w)
    # G2 exp_se3
    w = torch.nn.functional.normalize(torch.randn(8, 3, generator=g), dim=-1)
    S = torch.cat([w, torch.randn(8, 3, generator=g)], -1)
    theta = torch.tensor([[1e-4], [0.01], [0.3], [1.0], [2.0], [3.0], [0.7], [1.3]])
    d["se3_S"], d["se3_theta"], d["se3_out"] = S, theta, U.exp_se3(S, theta)
    # G3 embedders
    x3, t1 = torch.randn(5, 3, generator=g), torch.rand(5, 1, generator=g)
    e3, n3 = U.get_embedder(10, 3)
    e1, n1 = U.get_embedder(6, 1)
    e1b, n1b = U.get_embedder(10, 1)
    d["emb_x3"], d["emb_x3_out"], d["emb_t1"], d["emb_t1_out"], d["emb_t1_10_out"] = x3, e3(x3), t1, e1(t1), e1b(t1)
    d["emb_dims"] = torch.tensor([n3, n1, n1b])
    # G4 SH <-> RGB, resize
    rgb = torch.rand(6, 3, generator=g)
    d["rgb"], d["rgb2sh"], d["sh2rgb"] = rgb, U.RGB2SH(rgb), U.SH2RGB(rgb)
    img = torch.rand(8, 12, 3, generator=g)
    d["img"], d["img_d2"], d["img_d4"] = img, U.resize_image(img, 2), U.resize_image(img, 4)
    # G5 bilinear_interp, non-integer and integer coordinates (the latter returns 0: recorded quirk)
    im = torch.arange(2 * 4 * 5 * 2, dtype=torch.float32).reshape(2, 4, 5, 2)
    bx = torch.tensor([[0.5, 1.25, 3.75, 2.0], [4.0, 0.0, 2.5, 1.0]])
    by = torch.tensor([[0.5, 2.5, 1.1, 1.0], [3.0, 0.0, 0.25, 2.0]])
    d["bil_img"], d["bil_x"], d["bil_y"], d["bil_out"] = im, bx, by, U.bilinear_interp(im, bx, by)
    v = torch.randn(3, 3, generator=g)
    d["hom_v"], d["hom_to"] = v, U.to_homogenous(v)
    d["hom_from"] = U.from_homogenous(torch.cat([v, torch.full((3, 1), 2.0)], -1))
    np.savez(os.path.join(OUT, "g_utils.npz"), **{k: t.numpy() for k, t in d.items()})
    print("g_utils.npz", len(d))


def fill_params(module):
    """Deterministic, construction-order-independent parameter values (the test applies the same
    function to the build's modules, so no state_dict needs to be stored)."""
    with torch.no_grad():
        for k, (name, p) in enumerate(sorted(module.state_dict().items())):
            n = p.numel()
            fan_in = p.shape[-1] if p.dim() > 1 else 256
            vals = torch.sin(torch.arange(n, dtype=torch.float64) * (0.37 + 0.011 * k) + k) / (fan_in**0.5)
            p.copy_(vals.reshape(p.shape).float())


def gen_mlp(U):
    src = slice_defs(os.path.join(REF, "freegaussian", "freegaussian_model.py"),
                     ["FreeGaussianDeformableModel", "FreeGaussianControllableModel"])  # fmt: skip
    ns = {"nn": torch.nn, "torch": torch, "F": torch.nn.functional, "get_embedder": U.get_embedder,
          "exp_se3": U.exp_se3}  # fmt: skip
    exec(compile(src, "<reference MLP slice>", "exec"), ns)
    d = {}
    g = torch.Generator().manual_seed(0)
    x = torch.rand(16, 3, generator=g) * 2 - 1
    for tag, kw in (("deform", dict()), ("deform_blender", dict(is_blender=True))):
        m = ns["FreeGaussianDeformableModel"](**kw)
        fill_params(m)
        d[f"{tag}.keys"] = torch.tensor([len(m.state_dict())])
        for ti, t in enumerate((0.0, 0.5, 1.0)):
            dx, rot, sc = m(x, torch.full((16, 1), t))
            d[f"{tag}.t{ti}.d_xyz"], d[f"{tag}.t{ti}.rot"], d[f"{tag}.t{ti}.scale"] = dx, rot, sc
    m = ns["FreeGaussianControllableModel"]()
    fill_params(m)
    val = torch.randn(16, 3, generator=g) * 0.1
    dx, rot, sc = m(x, val)
    d["control.value"], d["control.d_xyz"], d["control.rot"], d["control.scale"] = val, dx, rot, sc
    d["x"] = x
    np.savez_compressed(os.path.join(OUT, "g_mlp.npz"), **{k: t.detach().numpy() for k, t in d.items()})
    print("g_mlp.npz", len(d))


def gen_flow():
    from einops import rearrange
    from scipy.spatial.transform import Rotation as R

    src = slice_defs(os.path.join(REF, "preprocess", "epipolar_flow.py"), ["opengl2cv", "diff_2d_epipolar_flow"])

    def to4x4(p):
        return torch.cat([p, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=p.dtype)], 0)

    def inverse(p):  # nerfstudio.utils.poses.inverse on a [3,4] pose
        Rm, t = p[:3, :3], p[:3, 3:]
        return torch.cat([Rm.T, -Rm.T @ t], -1)

    def multiply(a, b):  # nerfstudio.utils.poses.multiply
        return torch.cat([a[:3, :3] @ b[:3, :3], a[:3, :3] @ b[:3, 3:] + a[:3, 3:]], -1)

    class Cam:
        def __init__(self, c2w, fx, fy, cx, cy, H, W):
            self.camera_to_worlds = c2w
            # float64 throughout: the reference multiplies A by a float64 velocity (epipolar_flow.py:270,309)
            self.fx, self.fy, self.cx, self.cy = (torch.tensor([v], dtype=torch.float64) for v in (fx, fy, cx, cy))
            self.H, self.W = H, W

        def get_image_coords(self, pixel_offset=0.5):
            yy, xx = torch.meshgrid(torch.arange(self.H), torch.arange(self.W), indexing="ij")
            return torch.stack([yy, xx], -1).double() + pixel_offset

    ns = {"torch": torch, "np": np, "rearrange": rearrange, "R": R, "to4x4": to4x4, "inverse": inverse,
          "multiply": multiply, "Cameras": object, "print": lambda *a, **k: None}  # fmt: skip
    exec(compile(src, "<reference flow slice>", "exec"), ns)
    H, W, fx, fy, cx, cy = 4, 6, 7.0, 9.0, 2.5, 1.5
    g = torch.Generator().manual_seed(0)
    Z = (torch.rand(H, W, 1, generator=g) * 3 + 1).double()
    Z[1, 2, 0] = float("inf")
    of = torch.randn(H, W, 2, generator=g).numpy()
    base = torch.cat([torch.linalg.qr(torch.randn(3, 3, generator=g)).Q, torch.randn(3, 1, generator=g)], -1)
    d = {"Z": Z, "opticalflow": torch.from_numpy(of), "K": torch.tensor([fx, fy, cx, cy]), "c2w0": base}
    # pair A: pure translation; pair B: pure rotation (about the camera's own axes)
    cA = base.clone()
    cA[:, 3] += torch.tensor([0.02, -0.01, 0.03])
    rot = torch.from_numpy(R.from_euler("xyz", [0.01, -0.02, 0.015]).as_matrix()).float()
    cB = base.clone()
    cB[:3, :3] = base[:3, :3] @ rot
    for tag, c1 in (("trans", cA), ("rot", cB)):
        out = ns["diff_2d_epipolar_flow"](Z, Cam(base, fx, fy, cx, cy, H, W), Cam(c1, fx, fy, cx, cy, H, W), of.copy())
        d[f"{tag}.c2w1"] = c1
        d[f"{tag}.sceneflow"] = torch.from_numpy(np.asarray(out["sceneflow"]))
        d[f"{tag}.interflow"] = torch.from_numpy(np.asarray(out["interflow"]))
    np.savez(os.path.join(OUT, "g_flow.npz"), **{k: t.numpy() for k, t in d.items()})
    print("g_flow.npz", len(d))


def _pose_stubs():
    def to4x4(p):
        return torch.cat([p, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=p.dtype)], 0)

    def inverse(p):  # nerfstudio.utils.poses.inverse on a [3,4] pose
        Rm, t = p[:3, :3], p[:3, 3:]
        return torch.cat([Rm.T, -Rm.T @ t], -1)

    def multiply(a, b):  # nerfstudio.utils.poses.multiply
        return torch.cat([a[:3, :3] @ b[:3, :3], a[:3, :3] @ b[:3, 3:] + a[:3, 3:]], -1)

    return to4x4, inverse, multiply


class _Cam:
    """The fields of nerfstudio's Cameras the two flow functions touch."""

    def __init__(self, c2w, fx, fy, cx, cy, H, W, dtype=torch.float64):
        self.camera_to_worlds = c2w
        self.fx, self.fy, self.cx, self.cy = (torch.tensor([v], dtype=dtype) for v in (fx, fy, cx, cy))
        self.H, self.W, self.dtype = H, W, dtype

    def get_image_coords(self, pixel_offset=0.5):
        yy, xx = torch.meshgrid(torch.arange(self.H), torch.arange(self.W), indexing="ij")
        return torch.stack([yy, xx], -1).to(self.dtype) + pixel_offset

    def get_intrinsics_matrices(self):
        K = torch.eye(3, dtype=torch.float32)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = float(self.fx), float(self.fy), float(self.cx), float(self.cy)
        return K


def flow_bp_pairs():
    """The camera pairs of g_flow_bp.npz (shared with the test): the two pairs of g_flow.npz and
    two small-motion pairs about an axis-aligned camera, where first-order theory applies."""
    from scipy.spatial.transform import Rotation as R

    g = torch.Generator().manual_seed(0)
    _ = torch.rand(4, 6, 1, generator=g), torch.randn(4, 6, 2, generator=g)  # keep g_flow.npz's stream
    base = torch.cat([torch.linalg.qr(torch.randn(3, 3, generator=g)).Q, torch.randn(3, 1, generator=g)], -1)
    cA = base.clone()
    cA[:, 3] += torch.tensor([0.02, -0.01, 0.03])
    rot = torch.from_numpy(R.from_euler("xyz", [0.01, -0.02, 0.015]).as_matrix()).float()
    cB = base.clone()
    cB[:3, :3] = base[:3, :3] @ rot
    ident = torch.cat([torch.eye(3), torch.zeros(3, 1)], -1)
    sT = ident.clone()
    sT[:, 3] = torch.tensor([2e-3, -1e-3, 3e-3])
    sR = ident.clone()
    sR[:3, :3] = torch.from_numpy(R.from_euler("xyz", [1e-3, -2e-3, 1.5e-3]).as_matrix()).float()
    return {"trans": (base, cA), "rot": (base, cB), "small_trans": (ident, sT), "small_rot": (ident, sR)}


def gen_flow_bp():
    from torch.linalg import inv

    to4x4, inverse, multiply = _pose_stubs()
    src = slice_defs(os.path.join(REF, "preprocess", "epipolar_flow_bp.py"), ["manual2cv", "diff_2d_epipolar_flow"])
    ns = {"torch": torch, "np": np, "inv": inv, "to4x4": to4x4, "inverse": inverse, "multiply": multiply,
          "Cameras": object}  # fmt: skip
    exec(compile(src, "<reference flow_bp slice>", "exec"), ns)
    H, W, fx, fy, cx, cy = 4, 6, 7.0, 9.0, 2.5, 1.5
    g = torch.Generator().manual_seed(3)
    Z = torch.rand(H, W, 1, generator=g) * 3 + 1
    Z1 = Z + torch.randn(H, W, 1, generator=g) * 0.01
    Zi = Z.clone()
    Zi[1, 2, 0] = float("inf")
    of = torch.randn(H, W, 2, generator=g).numpy()
    d = {"Z": Zi, "Z1": Z1, "opticalflow": torch.from_numpy(of), "K": torch.tensor([fx, fy, cx, cy])}
    for tag, (c0, c1) in flow_bp_pairs().items():
        cam0, cam1 = _Cam(c0, fx, fy, cx, cy, H, W, torch.float32), _Cam(c1, fx, fy, cx, cy, H, W, torch.float32)
        out = ns["diff_2d_epipolar_flow"](Zi, Z1, cam0, cam1, of.copy())
        d[f"{tag}.c2w0"], d[f"{tag}.c2w1"] = c0, c1
        d[f"{tag}.sceneflow"] = torch.from_numpy(np.asarray(out["sceneflow"]))
        d[f"{tag}.interflow"] = torch.from_numpy(np.asarray(out["interflow"]))
    np.savez(os.path.join(OUT, "g_flow_bp.npz"), **{k: t.numpy() for k, t in d.items()})
    print("g_flow_bp.npz", len(d))


def slice_methods(path, cls, names):
    """Source of the named methods of a reference class, dedented (executed, never saved)."""
    import textwrap

    src = open(path).read()
    lines = src.splitlines()
    out = []
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for m in node.body:
                if isinstance(m, ast.FunctionDef) and m.name in names:
                    first = min([m.lineno] + [d.lineno for d in m.decorator_list])
                    out.append(textwrap.dedent("\n".join(lines[first - 1 : m.end_lineno])))
    assert len(out) == len(names), (names, len(out))
    return "\n\n".join(out)


DENSIFY_NAMES = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")
# (tag, step, config overrides, n, sh coefficients stored - 1, special)
DENSIFY_CASES = [
    ("densify_screen", 3500, {}, 220, 3, None),  # split + dup + cull, screen-size tests on, too-big culling on
    ("densify_noscreen", 4500, {}, 220, 3, None),  # past stop_screen_size_at
    ("densify_early", 900, {"refine_start": 500}, 220, 3, None),  # before refine_every * reset_alpha_every
    ("cull_only", 15100, {}, 220, 3, None),  # past stop_split_at
    ("opacity_reset", 3100, {}, 220, 3, None),  # step % reset_interval == refine_every
    ("full_sh", 3500, {}, 90, 15, None),  # the shipped layout: 15 higher-order coefficients
    ("dup_only", 4500, {}, 150, 3, "dup_only"),  # no split at all, duplicates only
    ("quirk", 3500, {}, 64, 3, "quirk"),  # split AND duplicated: dups evaluated after the in-place shrink
]
DENSIFY_CFG = dict(refine_every=100, refine_start=500, stop_split_at=15000, reset_alpha_every=30, densify_grad_thresh=0.0008,
                   densify_size_thresh=0.01, n_split_samples=2, cull_alpha_thresh=0.1, cull_scale_thresh=0.5,
                   continue_cull_post_densification=True, cull_screen_size=0.15, split_screen_size=0.05,
                   stop_screen_size_at=4000)  # fmt: skip  (reference defaults, freegaussian_model.py:58-88)


def densify_inputs(tag, step, n, k_rest, special, seed):
    """Seeded parameters, Adam moments and statistics of one case (the fixture stores them too)."""
    g = torch.Generator().manual_seed(seed)
    p = {
        "means": torch.rand(n, 3, generator=g) * 2 - 1,
        # sizes straddle densify_size_thresh (0.01) AND its 1.6x band, a few beyond cull_scale_thresh
        "scales": torch.randn(n, 3, generator=g) * 0.9 - 4.6,
        "quats": torch.randn(n, 4, generator=g) * (1.0 + torch.rand(n, 1, generator=g)),
        "features_dc": torch.rand(n, 3, generator=g),
        "features_rest": torch.randn(n, k_rest, 3, generator=g) * 0.1,
        "opacities": torch.randn(n, 1, generator=g) * 2.0,  # straddle cull_alpha_thresh
    }
    p["scales"][:6] = 0.2
    stats = {
        "xys_grad_norm": torch.rand(n, generator=g) * 4e-5,
        "vis_counts": torch.randint(1, 5, (n,), generator=g).float(),
        "max_2Dsize": torch.rand(n, generator=g) * 0.2,
    }
    if special == "dup_only":
        p["scales"].fill_(-7.0)
        p["opacities"].fill_(2.0)
        stats["xys_grad_norm"] = (torch.arange(n) % 7 == 0).float()
        stats["vis_counts"] = torch.ones(n)
        stats["max_2Dsize"] = torch.zeros(n)
    if special == "quirk":
        p["scales"].fill_(-9.0)
        p["scales"][7] = torch.log(torch.tensor(0.013))
        p["opacities"].fill_(2.0)
        stats["xys_grad_norm"] = torch.zeros(n)
        stats["xys_grad_norm"][7] = 1.0
        stats["vis_counts"] = torch.ones(n)
        stats["max_2Dsize"] = torch.zeros(n)
    mom = {k: {"exp_avg": torch.randn(v.shape, generator=g) * 0.01, "exp_avg_sq": torch.rand(v.shape, generator=g) * 1e-4}
           for k, v in p.items()}  # fmt: skip
    return p, mom, stats


def gen_densify():
    """refinement_after & co. and after_train_iter run as methods of a stub ``self``.  What the stub
    supplies: config / step / statistics attributes, the parameter accessors, `CONSOLE`, and
    gsplat's `quat_to_rotmat` (absent here: the standard wxyz formula, oracle/densify_oracle.py) --
    everything else (thresholds, mask order, concatenation order, optimizer surgery, the
    split-then-duplicate quirk) is the reference's own code executing."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from oracle.densify_oracle import quat_to_rotmat

    names = ["remove_from_optim", "remove_from_all_optim", "dup_in_optim", "dup_in_all_optim", "after_train_iter",
             "refinement_after", "cull_gaussians", "split_gaussians", "dup_gaussians"]  # fmt: skip
    src = slice_methods(os.path.join(REF, "freegaussian", "freegaussian_model.py"), "FreeGaussianModel", names)
    from typing import Optional

    drawn = []

    class TorchTap:
        """`torch` as the slice sees it: everything passes through; randn draws are recorded so the
        fixture does not depend on this torch build's RNG stream."""

        def __getattr__(self, name):
            return getattr(torch, name)

        def randn(self, *a, **k):
            out = torch.randn(*a, **k)
            drawn.append(out.clone())
            return out

    ns = {"torch": TorchTap(), "Optional": Optional, "Optimizers": object, "quat_to_rotmat": quat_to_rotmat,
          "CONSOLE": types.SimpleNamespace(log=lambda *a, **k: None)}  # fmt: skip
    exec(compile(src, "<reference densify slice>", "exec"), ns)

    class Stub:
        device = torch.device("cpu")
        num_points = property(lambda self: self.gauss_params["means"].shape[0])

        def get_gaussian_param_groups(self):
            return {k: [self.gauss_params[k]] for k in DENSIFY_NAMES}

    for k in DENSIFY_NAMES:
        setattr(Stub, k, property(lambda self, k=k: self.gauss_params[k]))
    for nme in names:
        setattr(Stub, nme, ns[nme])

    d = {}
    for ci, (tag, step, over, n, k_rest, special) in enumerate(DENSIFY_CASES):
        p, mom, stats = densify_inputs(tag, step, n, k_rest, special, seed=100 + ci)
        m = Stub()
        m.config = types.SimpleNamespace(**{**DENSIFY_CFG, **over})
        m.step, m.num_train_data, m.last_size = step, 60, (96, 160)
        m.gauss_params = torch.nn.ParameterDict({k: torch.nn.Parameter(v.clone()) for k, v in p.items()})
        m.xys_grad_norm, m.vis_counts, m.max_2Dsize = (stats[k].clone() for k in ("xys_grad_norm", "vis_counts", "max_2Dsize"))
        opts = {}
        for k in DENSIFY_NAMES:
            o = torch.optim.Adam([m.gauss_params[k]], lr=1e-3)
            o.state[m.gauss_params[k]] = {"step": torch.tensor(1.0), "exp_avg": mom[k]["exp_avg"].clone(),
                                          "exp_avg_sq": mom[k]["exp_avg_sq"].clone()}  # fmt: skip
            opts[k] = o
        torch.manual_seed(1234 + ci)
        drawn.clear()
        m.refinement_after(types.SimpleNamespace(optimizers=opts), step)
        assert len(drawn) <= 1  # the one draw of split_gaussians (:530)
        d[f"{tag}.samples"] = drawn[0] if drawn else torch.zeros(0, 3)
        n_out = m.num_points
        for k in DENSIFY_NAMES:
            d[f"{tag}.in.{k}"] = p[k]
            d[f"{tag}.out.{k}"] = m.gauss_params[k].detach()
            st = opts[k].state[opts[k].param_groups[0]["params"][0]]
            assert opts[k].param_groups[0]["params"][0] is m.gauss_params[k] and len(opts[k].state) == 1
            for mm in ("exp_avg", "exp_avg_sq"):
                d[f"{tag}.in.{k}.{mm}"] = mom[k][mm]
                d[f"{tag}.out.{k}.{mm}"] = st[mm]
        for k, v in stats.items():
            d[f"{tag}.in.{k}"] = v
        d[f"{tag}.meta"] = torch.tensor([step, n, n_out, 1234 + ci])
        assert m.xys_grad_norm is None and m.max_2Dsize is None
        print(" ", tag, n, "->", n_out)

    # S1: after_train_iter over two steps (:369-392)
    g = torch.Generator().manual_seed(77)
    n = 40
    m = Stub()
    m.config = types.SimpleNamespace(**DENSIFY_CFG)
    m.gauss_params = {"means": torch.zeros(n, 3)}
    m.xys_grad_norm = m.vis_counts = m.max_2Dsize = None
    m.last_size = (96, 160)
    for it in range(2):
        m.step = 700 + it
        m.radii = (torch.rand(n, generator=g) * 30 - 8).clamp_min(0).to(torch.int32)
        m.xys = types.SimpleNamespace(absgrad=torch.randn(1, n, 2, generator=g))
        d[f"s1.radii{it}"], d[f"s1.absgrad{it}"] = m.radii.clone(), m.xys.absgrad.clone()
        m.after_train_iter(m.step)
        d[f"s1.xys_grad_norm{it}"], d[f"s1.vis_counts{it}"], d[f"s1.max_2Dsize{it}"] = (
            m.xys_grad_norm.clone(), m.vis_counts.clone(), m.max_2Dsize.clone())  # fmt: skip
    np.savez_compressed(os.path.join(OUT, "g_densify.npz"), **{k: t.numpy() for k, t in d.items()})
    print("g_densify.npz", len(d))


def gen_flow_query(U):
    """query_3d_gaussian_flow / _grid (freegaussian_model.py:662-751, dead code upstream) as methods
    of a stub; `inverse` / `to4x4` are the batched nerfstudio pose helpers they call."""
    from torch.linalg import inv

    def inverse(p):
        R, t = p[..., :3, :3], p[..., :3, 3:]
        return torch.cat([R.transpose(-1, -2), -R.transpose(-1, -2) @ t], -1)

    def to4x4(p):
        bottom = torch.zeros_like(p[..., :1, :])
        bottom[..., 0, 3] = 1.0
        return torch.cat([p, bottom], -2)

    src = slice_methods(os.path.join(REF, "freegaussian", "freegaussian_model.py"), "FreeGaussianModel",
                        ["query_3d_gaussian_flow", "query_3d_gaussian_flow_grid"])  # fmt: skip
    ns = {"torch": torch, "inv": inv, "inverse": inverse, "to4x4": to4x4, "bilinear_interp": U.bilinear_interp}
    exec(compile(src, "<reference flow query slice>", "exec"), ns)
    g = torch.Generator().manual_seed(11)
    H, W, N = 24, 40, 30
    means2d = torch.rand(1, N, 2, generator=g) * torch.tensor([W + 8.0, H + 8.0]) - 4.0  # some off screen
    means2d[0, 3] = torch.tensor([7.0, 5.0])  # integer coordinates: the bilinear quirk gives 0 there
    Z0 = torch.rand(1, H, W, 1, generator=g) * 3 + 1
    interflow = torch.randn(1, H, W, 2, generator=g) * 1.5
    c2w1 = torch.cat([torch.linalg.qr(torch.randn(3, 3, generator=g)).Q, torch.randn(3, 1, generator=g)], -1)[None]
    K = torch.tensor([[30.0, 0.0, 19.5], [0.0, 28.0, 11.5], [0.0, 0.0, 1.0]])
    d = {"means2d": means2d, "Z0": Z0, "interflow": interflow, "c2w1": c2w1, "K": K}
    d["plain"] = ns["query_3d_gaussian_flow"](None, means2d, Z0, interflow, c2w1, K)["p1_3d2"]
    d["grid_16_8"] = ns["query_3d_gaussian_flow_grid"](None, means2d, Z0, interflow, c2w1, K)["p1_3d2"]
    d["grid_8_2"] = ns["query_3d_gaussian_flow_grid"](None, means2d, Z0, interflow, c2w1, K, 8, 2)["p1_3d2"]
    np.savez(os.path.join(OUT, "g_flow_query.npz"), **{k: t.numpy() for k, t in d.items()})
    print("g_flow_query.npz", {k: tuple(v.shape) for k, v in d.items() if k in ("plain", "grid_16_8", "grid_8_2")})


if __name__ == "__main__":
    U = load_utils()
    gen_utils(U)
    gen_mlp(U)
    gen_flow()
    gen_flow_bp()
    gen_densify()
    gen_flow_query(U)
