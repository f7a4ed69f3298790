"""What the forward's walk report (fg_raster_jobs_fwd walk_out: the entries some strip walked, when beyond 2560) says on scenes
where heavy tiles pay and where they do not.  Usage: python scripts/walk_values.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from policy_regret import build_scene, measure  # noqa: E402

from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import apply_layout, load_trained_scene, synthetic_scene  # noqa: E402

CASES = {
    "bench clustered:0.8:0.2": lambda: apply_layout(synthetic_scene(1_000_000, 1920, 1080, n_views=8, sh_degree=3, seed=42), "clustered:0.8:0.2"),
    "seed 11: clustered:0.6:0.4, small half-transparent": lambda: build_scene(dict(n=1_000_000, cam_radius=4.0, scale_mean=0.005, scale_max=0.05, opac_std=0.5, opac_shift=0.0, layout="clustered:0.6:0.4")),
    "seed 11: clustered:0.8:0.4+needles:0.5:10": lambda: build_scene(dict(n=1_000_000, cam_radius=4.0, scale_mean=0.005, scale_max=0.05, opac_std=0.5, opac_shift=0.0, layout="clustered:0.8:0.4+needles:0.5:10")),
    "the gate's faint cluster (60 000 splats of 4 % opacity in a ball of 0.2, 30 000 around)": lambda: _faint(),
    "trained": lambda: load_trained_scene(os.path.join(ROOT, "data", "trained_scene_r06.npz")),
}


def _faint():
    import math

    sc = synthetic_scene(90_000, 1920, 1080, n_views=8, sh_degree=3, seed=5, log_scale_mean=math.log(0.03))
    sc.means[:60_000] = sc.means[:60_000] * 0.1 + torch.tensor([0.8, -0.2, 0.0])
    sc.opacities[:60_000] *= 0.04
    return sc


def main():
    dev = torch.device("cuda", 0)
    for name, make in CASES.items():
        try:
            sc = make()
        except Exception as e:  # noqa: BLE001
            print(name, "skipped:", repr(e)[:100])
            continue
        out = {}
        for mode in ("always", "never"):
            ctx = ops.RasterContext(env={"FG_HEAVY_TILES": mode})
            g = [getattr(sc, k).to(dev).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
            vms, Ks = sc.viewmats.to(dev), sc.Ks.to(dev)
            vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(1)).to(dev)
            walks, longest = [], []
            import time
            ts = []
            for i in range(48):
                for t in g:
                    t.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with ops.use(ctx):
                    r, _, info = rasterization(*g, vms[i % 8 : i % 8 + 1], Ks[i % 8 : i % 8 + 1], sc.width, sc.height, sh_degree=sc.sh_degree,
                                               render_mode="RGB", packed=False, absgrad=True)
                    r.backward(vr)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
                walks.append(ctx.last_walk)
                offs = info["raster_isect_offsets"].reshape(-1)
                longest.append(int((offs[1:] - offs[:-1]).max()))
            ts = sorted(ts[16:])
            out[mode] = (ts[len(ts) // 2], walks[16:24], longest[16:24])
            ctx.release_workspaces()
        print(f"{name}: median step always {out['always'][0]:.3f} ms / never {out['never'][0]:.3f} ms; reported walks over eight views (always) "
              f"{out['always'][1]} (never) {out['never'][1]}; longest raster lists {out['never'][2]}", flush=True)


if __name__ == "__main__":
    main()
