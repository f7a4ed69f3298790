"""Whole 1920x1080 frame of EVERY view of the bench scene against the oracle (torch front end + C compositor):
gradient relative L2 per parameter, per view -- bench.py checks only the view its timed region happens to end on.
Usage (GPU box): python scripts/full_frame_views.py [n_gauss]      (FG_RASTER_LIB=<other build> for an A/B)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
scene = synthetic_scene(n, 1920, 1080, n_views=bench.N_VIEWS, sh_degree=3, seed=42)  # the bench scene
worst = {}
for view in range(bench.N_VIEWS):
    r = bench.cpu_full_frame(scene, view, 3)
    g = r["grad_rel_l2_hip_vs_oracle"]
    print(json.dumps({"view": view, "psnr_db": round(r["psnr_hip_vs_oracle_db"], 1), "reference_lists_bit_exact": r["reference_lists_bit_exact"],
                      **{k: float(f"{v:.3g}") for k, v in g.items()}}), flush=True)
    for k, v in g.items():
        worst[k] = max(worst.get(k, 0.0), v)
print(json.dumps({"worst_over_views": {k: float(f"{v:.3g}") for k, v in worst.items()}, "bar": 1e-4,
                  "lib": os.environ.get("FG_RASTER_LIB", "libfgraster.so")}))
