"""What the factored view-DP exchange of the TRAINING path (viewdp.ModelViewDP) would put on the links per rank, on the
bench scene and its clustered / needle layouts at 1M Gaussians / 1920x1080: per view the Gaussians that are visible
(radii > 0), the ones that carry a colour gradient (took part in a pixel), and the gathered block's size in its dense
and its sparse form.  One process, no collective: the numbers every rank would see in `dp.bytes_last_step`.
Usage: python scripts/dp_payload_bytes.py [layout ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import harness, viewdp  # noqa: E402
from freegaussian_amd.model import Camera, FreeGaussianModel, FreeGaussianModelConfig  # noqa: E402
from freegaussian_amd.scenes import apply_layout, synthetic_scene  # noqa: E402

layouts = sys.argv[1:] or ["uniform", "clustered:0.5:0.4", "clustered:0.8:0.2", "needles:0.3:10"]
dev = torch.device("cuda", 0)
W, H, n, N_VIEWS = 1920, 1080, 1_000_000, 8
out = {}
for layout in layouts:
    sc = apply_layout(synthetic_scene(n, W, H, n_views=N_VIEWS, sh_degree=3, seed=42), layout)
    cfg = FreeGaussianModelConfig(background_color="random", num_downscales=0, warm_up=10**9)
    model = FreeGaussianModel(cfg, seed_points=sc.means, init_scales=-4.0)
    with torch.no_grad():
        gp = model.gauss_params
        gp["scales"].copy_(sc.scales.log())
        gp["quats"].copy_(sc.quats)
        gp["opacities"].copy_(torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6))[:, None])
        gp["features_dc"].copy_(sc.colors[:, 0])
        gp["features_rest"].copy_(sc.colors[:, 1:])
    model = model.to(dev).train()
    model.step_cb(3001)
    dp = viewdp.ModelViewDP(model, sparse="auto")
    rows, vis = [], []
    for v in range(N_VIEWS):
        c2w = torch.linalg.inv(sc.viewmats[v])
        c2w[:3, 1:3] *= -1
        K = sc.Ks[v]
        cam = Camera(c2w[None, :3], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), W, H, times=torch.tensor([[0.0]]))
        for p in model.parameters():
            p.grad = None
        with dp.step():
            o = model.get_outputs(cam)
            harness.main_loss(o["rgb"], torch.rand(H, W, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(v))).backward()
        rows.append(dp.bytes_last_step["rows_with_colour_gradient"][0])
        vis.append(int((model.radii > 0).sum()))
    b = dp.bytes_last_step
    cap = int(max(rows) * 1.25) + 1024
    dense, sparse = b["dense_block_bytes"], (4 + cap * 4) * 4
    out[layout] = {"gaussians": n, "visible_per_view": [min(vis), max(vis)], "rows_with_colour_gradient_per_view": [min(rows), max(rows)],
                   "fraction_with_colour_gradient": round(max(rows) / n, 4), "dense_block_bytes": dense,
                   "sparse_block_bytes_capacity_1.25x": sparse, "auto_picks": "sparse" if sparse < 0.85 * dense else "dense",
                   "received_per_rank_at_8_ranks_MB": {"dense": round(7 * dense / 1e6, 1), "sparse": round(7 * sparse / 1e6, 1)},
                   "all_reduce_bytes": b["all_reduce"], "plain_all_reduce_would_be": b["plain_all_reduce_would_be"]}
    del model, dp
    torch.cuda.empty_cache()
print(json.dumps(out))
