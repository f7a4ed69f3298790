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


if __name__ == "__main__":
    U = load_utils()
    gen_utils(U)
    gen_mlp(U)
    gen_flow()
