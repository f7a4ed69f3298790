"""Randomised HIP-vs-oracle parity cases, shared by ``tests/test_gpu_parity.py`` (the gate) and ``scripts/fuzz_parity.py``
(longer sweeps): random sizes, image shapes, SH degrees, render modes, raster modes, packed / unpacked, with hostile
Gaussians mixed in (behind the camera, 25 x scale, sub-pixel, below the alpha threshold, unnormalised quaternions).

The bar (tests/helpers.py): integers ``torch.equal``; images through ``close_except_knife_edge`` at ``REL_TOL`` /
``KNIFE_EDGE_MAX_FRAC``; every gradient's relative L2 below ``REL_TOL``.  ONE escape for a gradient: arbitration by an fp64
run of the oracle.  The gradient is linear in the cotangents, so K draws of (v_render, v_alpha) on the same forward give K
samples of each fp32 implementation's distance from the fp64 gradient; the HIP path passes iff

    rel_K(HIP) <= 1.5 * max(rel_K(fp32 oracle), REL_TOL),
    rel_K(x) = sqrt(sum_k |g_x,k - g_fp64,k|^2) / sqrt(sum_k |g_fp64,k|^2)   (the K draws' gradients as ONE stacked vector)

and both numbers are recorded (``FG_PARITY_REPORT``).  (Pooled, not a mean of per-draw ratios: a draw whose fp64 gradient
happens to be small has a huge ratio on both sides and would decide the comparison by itself.)  Why K draws and not the one the case was caught with: the cases
that need arbitration are one- or two-Gaussian scenes rendered as expected depth, where d = D / alpha_out is constant over
the splat and its gradient is a sum of per-pixel rounding residues of ``alpha_out = 1 - T`` (ulp(1) / alpha each); a single
draw of that sum lands anywhere between 0.01 and 3 times its RMS -- for EITHER implementation (profiles/r05_ed_outlier.md:
seed 11 case 1, the fp32 oracle's own draw was at 0.06 of its RMS, the HIP path's at 2.2; over 24 draws 1.22e-3 / 1.47e-3).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import helpers  # noqa: E402
from helpers import KNIFE_EDGE_MAX_FRAC, REL_TOL, _rel_l2, close_except_knife_edge  # noqa: E402

NAMES = ("means", "quats", "scales", "opacities", "colors")
ARBITER_FACTOR = 1.5


class Case:
    """The inputs of case ``case`` of sweep ``seed0`` (the generator of rounds 1-4's scripts/fuzz_parity.py, draw for
    draw: the committed sweeps of profiles/r0*_fuzz_parity.txt are the same cases)."""

    def __init__(self, seed0: int, case: int, big: bool = False):
        from freegaussian_amd.scenes import synthetic_scene

        self.seed0, self.case, self.big = seed0, case, big
        g = self.g = torch.Generator().manual_seed(seed0 * 1000 + case)
        ri = self.ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
        n = [1, 2, 17, 300, 3000, 12000][ri(0, 5)]
        W, H = ri(17, 300), ri(17, 200)
        if big:
            n = [3000, 12000, 40000][ri(0, 2)]
            W, H = ri(640, 1920), ri(360, 1080)
        self.n, self.W, self.H = n, W, H
        self.deg = [None, 0, 1, 2, 3][ri(0, 4)]
        self.mode = ["RGB", "RGB+ED", "ED"][ri(0, 2)]
        self.rmode = ["classic", "antialiased"][ri(0, 1)]
        self.packed = bool(ri(0, 1))
        sc = self.sc = synthetic_scene(n, W, H, n_views=2, seed=seed0 * 1000 + case)
        k = max(1, n // 10)
        with torch.no_grad():  # hostile rows
            sc.means[:k] *= 5.0  # far off / behind
            sc.scales[k : 2 * k] *= 25.0  # huge
            sc.scales[2 * k : 3 * k] *= 0.02  # sub-pixel
            sc.opacities[3 * k : 4 * k] = 0.003  # below the alpha skip almost everywhere
            sc.quats[4 * k : 5 * k] *= 7.0  # unnormalised
        self.colors = sc.colors if self.deg is not None else torch.sigmoid(sc.colors[:, 0, :])
        self.view = ri(0, 1)
        self.kw = dict(sh_degree=self.deg, render_mode=self.mode, packed=self.packed, absgrad=True, rasterize_mode=self.rmode)

    def __str__(self):
        return (f"seed {self.seed0} case {self.case:3d} n={self.n:5d} {self.W}x{self.H} sh={self.deg} {self.mode:6s} "
                f"{self.rmode:11s} packed={int(self.packed)}")

    def leaves(self, dtype=torch.float32, device="cpu"):
        sc = self.sc
        return [t.detach().to(device=device, dtype=dtype).clone().requires_grad_(True)
                for t in (sc.means, sc.quats, sc.scales, sc.opacities, self.colors)]  # fmt: skip

    def cameras(self, dtype=torch.float32, device="cpu"):
        v = self.view
        return (self.sc.viewmats[v : v + 1].to(device=device, dtype=dtype), self.sc.Ks[v : v + 1].to(device=device, dtype=dtype))


def _backward(ins, r, a, vr, va, retain=True):
    for t in ins:
        t.grad = None
    if r.requires_grad or a.requires_grad:
        ((r * vr.to(r)).sum() + (a * va.to(a)).sum()).backward(retain_graph=retain)
    return [None if t.grad is None else t.grad.detach().double().cpu().clone() for t in ins]


def _grad_errors(g_hip, g_ref):
    errs = []
    for x1, x0 in zip(g_hip, g_ref):
        if x0 is None or float(x0.abs().max()) == 0.0:
            errs.append(0.0 if (x1 is None or float(x1.abs().max()) == 0.0) else 1.0)
        else:
            errs.append(_rel_l2(x1, x0))
    return errs


def arbitrate(case: Case, ins32, r32, a32, ins_hip, r_hip, a_hip, inputs, draws: int):
    """Relative L2 distance to the fp64 oracle's gradient pooled over ``draws`` cotangent draws (module docstring), for the
    inputs listed: -> {input index: (the fp32 oracle's, the HIP path's)}."""
    from oracle import raster_oracle as O

    ins64 = case.leaves(torch.float64)
    r64, a64, _ = O.rasterization(*ins64, *case.cameras(torch.float64), case.W, case.H, **case.kw)
    gd = torch.Generator().manual_seed(77_000 + 1000 * case.seed0 + case.case)
    sq = {j: [0.0, 0.0, 0.0] for j in inputs}
    for _ in range(draws):
        vr = torch.randn(r64.shape, generator=gd)
        va = torch.randn(a64.shape, generator=gd)
        g64 = _backward(ins64, r64, a64, vr.double(), va.double())
        g32 = _backward(ins32, r32, a32, vr, va)
        gh = _backward(ins_hip, r_hip, a_hip, vr, va)
        for j in inputs:
            sq[j][0] += float((g32[j] - g64[j]).norm()) ** 2
            sq[j][1] += float((gh[j] - g64[j]).norm()) ** 2
            sq[j][2] += float(g64[j].norm()) ** 2
    return {j: ((sq[j][0] / max(sq[j][2], 1e-60)) ** 0.5, (sq[j][1] / max(sq[j][2], 1e-60)) ** 0.5) for j in inputs}


def check(case: Case, device="cuda", draws: int = 32, log=print):
    """Run the case through the oracle and (twice: the second call of a shape may take the one-call-per-direction path)
    through the HIP path; -> (ok, message).  Comparisons are recorded through tests/helpers.py."""
    from freegaussian_amd import rasterization
    from oracle import raster_oracle as O

    kw = case.kw
    oracle_kw = kw
    if case.big:  # the C restatement composites (the torch compositing costs ~10 ms per tile); same backward semantics
        from oracle import c_oracle as CO

        oracle_kw = dict(kw, compositor=CO.composite)
    ins0 = case.leaves()
    r0, a0, i0 = O.rasterization(*ins0, *case.cameras(), case.W, case.H, **oracle_kw)
    vr = torch.randn(r0.shape, generator=case.g)
    va = torch.randn(a0.shape, generator=case.g)
    g0 = _backward(ins0, r0, a0, vr, va)
    problems, notes = [], []
    worst = 0.0
    for attempt in range(2):
        ins1 = case.leaves(device=device)
        r1, a1, i1 = rasterization(*ins1, *case.cameras(device=device), case.W, case.H, **kw)
        if r1.shape != r0.shape or a1.shape != a0.shape:
            return False, f"shapes differ: {tuple(r1.shape)} / {tuple(r0.shape)}"
        for key in ("radii", "flatten_ids", "isect_offsets"):
            if not torch.equal(i1[key].cpu(), i0[key]):
                problems.append(f"{key} differ (call {attempt})")
        if case.packed and not torch.equal(i1["gaussian_ids"].cpu(), i0["gaussian_ids"]):
            problems.append(f"gaussian_ids differ (call {attempt})")
        if not close_except_knife_edge(r1, r0, REL_TOL, KNIFE_EDGE_MAX_FRAC):
            problems.append(f"render beyond the bar (call {attempt})")
        if not close_except_knife_edge(a1, a0, REL_TOL, KNIFE_EDGE_MAX_FRAC):
            problems.append(f"alpha beyond the bar (call {attempt})")
        g1 = _backward(ins1, r1, a1, vr, va)
        errs = _grad_errors(g1, g0)
        for e in errs:
            helpers._record("rel_l2", e)
        worst = max(worst, max(errs))
        over = [j for j, e in enumerate(errs) if e >= REL_TOL]
        if 1.0 in [errs[j] for j in over]:
            problems.append(f"a gradient is zero on one side only (call {attempt})")
            over = [j for j in over if errs[j] != 1.0]
        if over:
            if case.big:
                # (arbitration needs the torch compositing: the forward graphs of both oracles, at full size)
                ins0t = case.leaves()
                r0t, a0t, _ = O.rasterization(*ins0t, *case.cameras(), case.W, case.H, **kw)
            else:
                ins0t, r0t, a0t = ins0, r0, a0
            res = arbitrate(case, ins0t, r0t, a0t, ins1, r1, a1, over, max(2, draws // 4) if case.big else draws)
            for j in over:
                rms_or, rms_hip = res[j]
                helpers._record("arbiter_single_draw_hip_vs_oracle32", errs[j])
                helpers._record("arbiter_oracle32_vs_fp64_pooled", rms_or)
                helpers._record("arbiter_hip_vs_fp64_pooled", rms_hip)
                verdict = rms_hip <= ARBITER_FACTOR * max(rms_or, REL_TOL)
                notes.append(f"[{NAMES[j]}: {errs[j]:.1e} -> fp64 arbiter, pooled over draws: oracle32 {rms_or:.1e}, HIP {rms_hip:.1e}"
                             f"{'' if verdict else ' FAILS'}]")
                if not verdict:
                    problems.append(f"{NAMES[j]} gradient {errs[j]:.1e} and HIP further from fp64 than {ARBITER_FACTOR} x the fp32 oracle")
    msg = f"{case} I={i0['flatten_ids'].numel():7d} grad rel {worst:.1e} {'ok' if not problems else 'MISMATCH ' + '; '.join(problems)} {' '.join(notes)}"
    return not problems, msg
