"""Per VIEW of a scene: the longest supertile segment and the step time with the long-segment sort always / never (the flag is per
shape with a cooldown of 64 calls, so on a scene whose views differ it is on for all of them or for none).
Usage: python scripts/long_mode_by_view.py [trained | <layout> [n] [cam_radius] [scale_mean] [opac_shift] [opac_std]]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from policy_regret import build_scene  # noqa: E402

from freegaussian_amd import ops, rasterization  # noqa: E402
from freegaussian_amd.scenes import load_trained_scene  # noqa: E402


def main():
    a = sys.argv[1:]
    if not a or a[0] == "trained":
        sc = load_trained_scene(os.path.join(ROOT, "data", "trained_scene_r06.npz"))
    else:
        sc = build_scene(dict(layout=a[0], n=int(a[1]) if len(a) > 1 else 1_000_000, cam_radius=float(a[2]) if len(a) > 2 else 4.0,
                              scale_mean=float(a[3]) if len(a) > 3 else 0.01, scale_max=0.05, opac_shift=float(a[4]) if len(a) > 4 else 0.0,
                              opac_std=float(a[5]) if len(a) > 5 else 1.5))
    dev = torch.device("cuda", 0)
    g = [getattr(sc, k).to(dev).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
    vms, Ks = sc.viewmats.to(dev), sc.Ks.to(dev)
    vr = torch.randn(1, sc.height, sc.width, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    rows = []
    for v in range(vms.shape[0]):
        row = {"view": v}
        for mode in ("always", "never"):
            ctx = ops.RasterContext(env={"FG_LONG_SEGMENTS": mode})
            ts = []
            for i in range(40):
                for t in g:
                    t.grad = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with ops.use(ctx):
                    r, _, info = rasterization(*g, vms[v : v + 1], Ks[v : v + 1], sc.width, sc.height, sh_degree=sc.sh_degree, render_mode="RGB",
                                               packed=False, absgrad=True)
                    r.backward(vr)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = sorted(ts[12:])
            row[mode] = round(ts[len(ts) // 2], 4)
            row["longest_segment"] = int(ctx.longest_segment_seen)
            row["I"] = int(info["raster_flatten_ids"].numel())
            ctx.release_workspaces()
        row["always/never"] = round(row["always"] / row["never"], 3)
        rows.append(row)
        print(row, flush=True)


if __name__ == "__main__":
    main()
