"""Round-5 investigation of the fuzz sweep's outlier (seed 11, case 1: one visible Gaussian, `ED`, antialiased, packed):
the HIP path's opacity gradient was 9x further from an fp64 run of the oracle than the fp32 oracle.  Which stage?
Prints, per input, the relative L2 distance to the fp64 oracle of (a) the fp32 oracle, (b) the HIP path, (c) the HIP path
with the expected-depth division and its backward done by torch on the CPU, (d) the HIP compositing fed the ORACLE's
projected inputs and cotangents (raster backward alone), and the forward images' agreement."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import _rel_l2 as rel_l2  # noqa: E402
from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402
from oracle import raster_oracle as O  # noqa: E402

torch.set_num_threads(16)
seed0, case = int(sys.argv[1]) if len(sys.argv) > 1 else 11, int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator().manual_seed(seed0 * 1000 + case)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
n = [1, 2, 17, 300, 3000, 12000][ri(0, 5)]
W, H = ri(17, 300), ri(17, 200)
deg = [None, 0, 1, 2, 3][ri(0, 4)]
mode = ["RGB", "RGB+ED", "ED"][ri(0, 2)]
rmode = ["classic", "antialiased"][ri(0, 1)]
packed = bool(ri(0, 1))
sc = synthetic_scene(n, W, H, n_views=2, seed=seed0 * 1000 + case)
k = max(1, n // 10)
with torch.no_grad():
    sc.means[:k] *= 5.0
    sc.scales[k : 2 * k] *= 25.0
    sc.scales[2 * k : 3 * k] *= 0.02
    sc.opacities[3 * k : 4 * k] = 0.003
    sc.quats[4 * k : 5 * k] *= 7.0
colors = sc.colors if deg is not None else torch.sigmoid(sc.colors[:, 0, :])
v = ri(0, 1)
print(f"n={n} {W}x{H} sh={deg} {mode} {rmode} packed={packed} view={v}")
dev = "cuda"
names = ["means", "quats", "scales", "opac", "colors"]


def leaves(dt, device="cpu"):
    return [t.detach().to(device=device, dtype=dt).clone().requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, colors)]


def cams(dt, device="cpu"):
    return sc.viewmats[v : v + 1].to(device=device, dtype=dt), sc.Ks[v : v + 1].to(device=device, dtype=dt)


kw = dict(sh_degree=deg, render_mode=mode, packed=packed, absgrad=True, rasterize_mode=rmode)
i64 = leaves(torch.float64)
r64, a64, _ = O.rasterization(*i64, *cams(torch.float64), W, H, **kw)
vr = torch.randn(r64.shape, generator=g)
va = torch.randn(a64.shape, generator=g)


def grads(ins, r, a, wr=1.0, wa=1.0):
    for t in ins:
        t.grad = None
    (wr * (r * vr.to(r)).sum() + wa * (a * va.to(a)).sum()).backward(retain_graph=True)
    return [None if t.grad is None else t.grad.detach().cpu().double().clone() for t in ins]


i32 = leaves(torch.float32)
r32, a32, info32 = O.rasterization(*i32, *cams(torch.float32), W, H, **kw)
ih = leaves(torch.float32, dev)
rh, ah, infoh = rasterization(*ih, *cams(torch.float32, dev), W, H, **kw)
print("forward: alpha equal bits", float((ah.cpu() == a32).float().mean()), "render rel", rel_l2(rh, r32), "alpha rel", rel_l2(ah, a32))
nz = a32 > 0
print("alpha_out over alive pixels: min", float(a32[nz].min()), "max", float(a32.max()), "pixels", int(nz.sum()))
for label, wr, wa in (("render+alpha", 1, 1), ("render only", 1, 0), ("alpha only", 0, 1)):
    g64 = grads(i64, r64, a64, wr, wa)
    g32 = grads(i32, r32, a32, wr, wa)
    gh = grads(ih, rh, ah, wr, wa)
    for j, nm in enumerate(names):
        if g64[j] is None or float(g64[j].abs().max()) == 0:
            continue
        print(f"{label:13s} {nm:7s} |g64| {g64[j].norm():.3e}  oracle32-64 {(g32[j] - g64[j]).norm():.2e}  hip-64 {(gh[j] - g64[j]).norm():.2e}  hip-oracle32 {(gh[j] - g32[j]).norm():.2e}")

# (c) accumulated depth "D" mode on the GPU, the division and its autograd on the CPU
if mode.endswith("ED"):
    kwd = dict(kw, render_mode=mode.replace("ED", "D"))
    ih2 = leaves(torch.float32, dev)
    rD, aD, _ = rasterization(*ih2, *cams(torch.float32, dev), W, H, **kwd)
    rDc, aDc = rD.cpu(), aD.cpu()  # (differentiable device copies)
    di = rDc.shape[-1] - 1
    d = rDc[..., di : di + 1] / aDc.clamp(min=1e-10)
    rc = torch.cat([rDc[..., :di], d], -1)
    gc = grads(ih2, rc, aDc)
    g64 = grads(i64, r64, a64)
    gh = grads(ih, rh, ah)
    g32 = grads(i32, r32, a32)
    for j, nm in enumerate(names):
        if g64[j] is None or float(g64[j].abs().max()) == 0:
            continue
        print(f"division on CPU  {nm:7s} hip-64 {(gc[j] - g64[j]).norm():.2e} (division on GPU {(gh[j] - g64[j]).norm():.2e}, oracle32 {(g32[j] - g64[j]).norm():.2e})")

# (d) the compositing alone: oracle's projected inputs and cotangents into fg_raster_fwd / fg_raster_bwd
with torch.no_grad():
    vm, K = cams(torch.float32)
    proj = O.project(*[t.detach() for t in i32[:3]], vm[0], K[0], W, H, 0.3, 0.01, 1e10, 0.0)
    opac = i32[3].detach() * (proj.compensations if rmode == "antialiased" else 1.0)
    feats = proj.depths[:, None] if mode == "ED" else None
if feats is not None:
    tw, th = (W + 15) // 16, (H + 15) // 16
    _, isect, flat = O.isect_tiles(proj.means2d, proj.radii, proj.depths, 16, tw, th)
    offs = O.isect_offsets(isect, tw * th)
    args = (proj.means2d, proj.conics, feats, opac)
    ren, alp, _ = O.rasterize(*args, W, H, 16, offs, flat)
    v_ren = (vr[0] / alp.clamp(min=1e-10)).float()
    v_alp = (va[0] - vr[0] * ren / alp.clamp(min=1e-10) ** 2 * (alp > 1e-10)).float()[..., 0]
    ref32 = O.rasterize_backward(*args, W, H, 16, offs, flat, v_ren, v_alp, alpha_out=alp)
    args64 = tuple(t.double() for t in args)
    ren64, alp64, _ = O.rasterize(*args64, W, H, 16, offs, flat)
    ref64 = O.rasterize_backward(*args64, W, H, 16, offs, flat, v_ren.double(), v_alp.double(), alpha_out=alp64)
    ga = [t.to(dev).requires_grad_(True) for t in args]
    rg, ag, _ = ops.rasterize_to_pixels(ga[0], ga[1], ga[2], ga[3], W, H, 16, offs.to(dev), flat.to(dev), absgrad=True)
    print("raster alone forward: alpha bits equal", float((ag.cpu() == alp).float().mean()), "render rel", rel_l2(rg, ren))
    ((rg * v_ren.to(dev)).sum() + (ag[..., 0] * v_alp.to(dev)).sum()).backward()
    for nm, gi, r32_, r64_ in (("v_xy", ga[0].grad, ref32[0], ref64[0]), ("v_conic", ga[1].grad, ref32[2], ref64[2]),
                               ("v_depth", ga[2].grad, ref32[3], ref64[3]), ("v_opac", ga[3].grad, ref32[4], ref64[4])):
        print(f"raster alone  {nm:8s} |ref64| {r64_.norm():.3e}  oracle32-64 {(r32_.double() - r64_).norm():.2e}  hip-64 {(gi.cpu().double() - r64_).norm():.2e}")

# (e) is the fp32 oracle's own distance from fp64 a stable yardstick?  The gradient is linear in the cotangents: K draws of
# (vr, va) on the same forward give K samples of each implementation's fp32 noise.  RMS over the draws, per input.
K_DRAWS = int(os.environ.get("ED_DRAWS", "24"))
acc = {nm: [[], [], []] for nm in names}
gd = torch.Generator().manual_seed(12345)
for k_ in range(K_DRAWS):
    vr = torch.randn(r64.shape, generator=gd)
    va = torch.randn(a64.shape, generator=gd)
    g64 = grads(i64, r64, a64)
    g32 = grads(i32, r32, a32)
    gh = grads(ih, rh, ah)
    for j, nm in enumerate(names):
        if g64[j] is None or float(g64[j].abs().max()) == 0:
            continue
        acc[nm][0].append(float((g32[j] - g64[j]).norm() / g64[j].norm()))
        acc[nm][1].append(float((gh[j] - g64[j]).norm() / g64[j].norm()))
        acc[nm][2].append(float((gh[j] - g32[j]).norm() / g64[j].norm()))
for nm in names:
    if acc[nm][0]:
        t = torch.tensor(acc[nm])
        print(f"{K_DRAWS} cotangent draws  {nm:7s} rel L2 to fp64: oracle32 rms {t[0].pow(2).mean().sqrt():.2e} (min {t[0].min():.1e} max {t[0].max():.1e})"
              f"  hip rms {t[1].pow(2).mean().sqrt():.2e} (min {t[1].min():.1e} max {t[1].max():.1e})  hip-oracle32 rms {t[2].pow(2).mean().sqrt():.2e}")
