"""Work counters of the raster kernels on the bench scene (needs `make -C freegaussian_amd/csrc stats`
and FG_RASTER_LIB=freegaussian_amd/libfgraster_stats.so).  Prints, per launch: list entries walked,
pixel slots evaluated / with a contributing lane, contributing lanes, reductions."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FG_RASTER_LIB", os.path.join(ROOT, "freegaussian_amd", "libfgraster_stats.so"))
from freegaussian_amd import _lib, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
if len(sys.argv) > 2:  # a trained_scene.npz of scripts/train_e2e.py (its own cameras); [3] = view
    from freegaussian_amd.scenes import load_trained_scene

    sc = load_trained_scene(sys.argv[2])
    v = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    sc.viewmats, sc.Ks = sc.viewmats[v : v + 1], sc.Ks[v : v + 1]
else:
    sc = synthetic_scene(n, 1920, 1080, n_views=8, sh_degree=3, seed=42)
dev = torch.device("cuda", 0)
ins = [t.to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
lib = _lib.load()
lib.fg_debug_raster_stats.restype = ctypes.c_int
lib.fg_debug_raster_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
lib.fg_debug_raster_stats(buf, 1)
r, a, info = rasterization(*ins, sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), sc.width, sc.height, sh_degree=3, absgrad=True)
r.backward(torch.randn_like(r))
torch.cuda.synchronize()
lib.fg_debug_raster_stats(buf, 1)
s = list(buf)
I = info["flatten_ids"].numel()
out = {
    "N": int(sc.means.shape[0]), "V": int((info["radii"] > 0).sum()), "I_raster": int(info["raster_flatten_ids"].numel()),
    "I": I,
    "bwd": {"entries_staged(n_used)": s[5], "entries_in_lists": s[6], "entries_walked(per wave)": s[0],
            "slots_evaluated": s[1], "slots_with_valid_lane": s[2], "valid_lanes": s[3], "reductions": s[4],
            "lanes_per_live_slot": s[3] / max(s[2], 1), "live_slots_per_walked_entry": s[2] / max(s[0], 1)},
    "fwd": {"entries_walked(per wave)": s[8], "slots_evaluated": s[9], "slots_with_valid_lane": s[10],
            "valid_lanes": s[11], "lanes_per_live_slot": s[11] / max(s[10], 1)},
}
print(json.dumps(out, indent=1))
