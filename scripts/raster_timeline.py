"""Job timeline of the mixed raster launches on the bench scene (needs `make -C freegaussian_amd/csrc timeline`).

Every job of raster_fwd_mixed_kernel / raster_bwd_mixed_kernel records its start and end on the 100 MHz
wall clock and the SIMD it ran on.  Printed per kernel: the launch span, the jobs by kind with their
durations, and -- in 20 time slices of the launch -- how many wavefronts were resident per SIMD and which
share of the chip's 1024 SIMDs held 0 / 1 / 2-3 / >= 4 of them (a SIMD needs ~4 to issue at full rate).
Usage: python scripts/raster_timeline.py [n_gauss] [out.json] [frac:extent]
(frac:extent: that fraction of the Gaussians pulled into a ball of that extent at the centre, scripts/clustered_check.py)"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FG_RASTER_LIB", os.path.join(ROOT, "freegaussian_amd", "libfgraster_timeline.so"))
from freegaussian_amd import _lib, rasterization  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = synthetic_scene(n, 1920, 1080, n_views=8, sh_degree=3, seed=42)
if len(sys.argv) > 3 and sys.argv[3].startswith("trained:"):  # trained:<trained_scene.npz of scripts/train_e2e.py>
    from freegaussian_amd.scenes import load_trained_scene

    sc = load_trained_scene(sys.argv[3].split(":", 1)[1])
elif len(sys.argv) > 3 and sys.argv[3].startswith("{"):  # a layout of scripts/policy_regret.py, as JSON
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from policy_regret import build_scene

    sc = build_scene(json.loads(sys.argv[3]))
elif len(sys.argv) > 3 and sys.argv[3]:
    frac, ball = (float(v) for v in sys.argv[3].split(":"))
    sc.means[: int(frac * n)] *= ball / 2.0
TL_VIEW = int(os.environ.get("FG_TL_VIEW", "4"))
dev = torch.device("cuda", 0)
ins = [t.to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
lib = _lib.load()
lib.fg_debug_raster_timeline.restype = ctypes.c_int
lib.fg_debug_raster_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
CAP = 1 << 17
buf = np.zeros((CAP, 6), dtype=np.uint64)


info = None


def step(view):
    global info
    r, a, info = rasterization(*ins, sc.viewmats[view:view + 1].to(dev), sc.Ks[view:view + 1].to(dev), sc.width, sc.height,
                               sh_degree=3, absgrad=True)
    r.backward(torch.randn_like(r))
    torch.cuda.synchronize()


for v in range(int(os.environ.get("FG_TL_WARM", "4"))):  # (the host's per-shape policy settles over its first calls: uneven after one, even after eight)
    step(v % sc.viewmats.shape[0])
lib.fg_debug_raster_timeline(buf.ctypes.data, CAP, 1)
step(TL_VIEW)
cnt = lib.fg_debug_raster_timeline(buf.ctypes.data, CAP, 1)
rec = buf[:min(cnt, CAP)]
t0, t1, hw, what = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64), rec[:, 2], rec[:, 3]
what_lo = what & ((1 << 40) - 1)
marks_us = np.stack([(rec[:, 4] & 0xFFFFFFFF), (rec[:, 4] >> 32), (rec[:, 5] & 0xFFFFFFFF), (rec[:, 5] >> 32)],
                    axis=1).astype(np.float64) / 100.0  # FG_TL_MARK(0..3): microseconds after the job's start
kernel = (what & 0xF).astype(int)
strip = ((what >> 4) & 0xF).astype(int) - 1
parts = ((what >> 12) & 0xF).astype(int)
hwid = (hw & 0xFFFFFFFF).astype(np.int64)
stage_us = (hw >> 36).astype(np.float64) / 100.0  # time spent staging batches (gather -> LDS)
prologue_us = (what >> 40).astype(np.float64) / 100.0  # job start -> first batch
xcc = ((hw >> 32) & 0xF).astype(np.int64)
simd = (hwid >> 4) & 3
cu = (hwid >> 8) & 0xF
sh_ = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd_key = (((xcc * 8 + se) * 2 + sh_) * 16 + cu) * 4 + simd
out = {"jobs_recorded": int(cnt)}
for kid, name in ((1, "raster_fwd_mixed"), (2, "raster_bwd_mixed")):
    m = kernel == kid
    if not m.any():
        continue
    a, b, sk = t0[m], t1[m], simd_key[m]
    lo, hi = a.min(), b.max()
    span_us = (hi - lo) / 100.0
    dur = (b - a) / 100.0
    kinds = {}
    if kid == 1:
        kind = np.where(strip[m] < 0, "whole tile", np.where(strip[m] >= 4, "two strips", "one strip"))
    else:
        kind = np.where(parts[m] > 1, "list share", np.where(strip[m] < 0, "whole tile", "strips"))
    for kk in np.unique(kind):
        d = dur[kind == kk]
        kinds[str(kk)] = {"jobs": int(d.size), "mean_us": float(d.mean()), "p50_us": float(np.median(d)),
                          "p95_us": float(np.percentile(d, 95)), "max_us": float(d.max()),
                          "sum_ms": float(d.sum() / 1e3),
                          "staging_share": float(stage_us[m][kind == kk].sum() / d.sum()),
                          "prologue_share": float(prologue_us[m][kind == kk].sum() / d.sum()),
                          "prologue_mean_us": float(prologue_us[m][kind == kk].mean()),
                          "marks_mean_us": [float(x) for x in marks_us[m][kind == kk].mean(axis=0)]}
    usimd = np.unique(sk)
    nsl = 20
    edges = np.linspace(lo, hi, nsl + 1)
    slices = []
    for i in range(nsl):
        e0, e1 = edges[i], edges[i + 1]
        ov = np.clip(np.minimum(b, e1) - np.maximum(a, e0), 0, None) / (e1 - e0)  # resident fraction per job
        per = np.zeros(usimd.size)
        np.add.at(per, np.searchsorted(usimd, sk), ov)
        allsimd = np.concatenate([per, np.zeros(max(0, 1024 - usimd.size))])
        slices.append({"t_us": round((e0 - lo) / 100.0, 1), "waves_per_simd": round(float(allsimd.mean()), 2),
                       "simds_idle": round(float((allsimd < 0.25).mean()), 3),
                       "simds_1": round(float(((allsimd >= 0.25) & (allsimd < 1.5)).mean()), 3),
                       "simds_2_3": round(float(((allsimd >= 1.5) & (allsimd < 3.5)).mean()), 3),
                       "simds_4plus": round(float((allsimd >= 3.5).mean()), 3)})
    # end of the last job per SIMD, relative to the launch end: how ragged is the finish
    last_end = np.array([b[sk == u].max() for u in usimd])
    first_start = np.array([a[sk == u].min() for u in usimd])
    out[name] = {"span_us": span_us, "jobs": int(m.sum()), "simds_used": int(usimd.size),
                 "job_wave_time_ms": float(dur.sum() / 1e3),
                 "mean_resident_waves_per_simd": float(dur.sum() / span_us / 1024),
                 "simd_finish_before_end_us": {"mean": float(((hi - last_end) / 100.0).mean()),
                                               "p50": float(np.median((hi - last_end) / 100.0)),
                                               "p90": float(np.percentile((hi - last_end) / 100.0, 90))},
                 "simd_first_start_us": {"mean": float(((first_start - lo) / 100.0).mean()),
                                         "max": float(((first_start - lo) / 100.0).max())},
                 "per_xcd_finish_us": {int(x): float((b[xcc[m] == x].max() - lo) / 100.0) for x in np.unique(xcc[m])},
                 "kinds": kinds, "slices": slices}
# the longest jobs of each launch: what they were and how long their tile's list is
offs = info["raster_isect_offsets"].reshape(-1).long().cpu().numpy()  # [T + 1]: the lists the launches walked
n_list = int(offs[-1])
lens = offs[1:] - offs[:-1]
out["lists"] = {"entries": n_list, "mean": float(lens.mean()), "p99": float(np.percentile(lens, 99)), "max": int(lens.max())}
tile_id = ((what >> 16) & 0xFFFFFF).astype(np.int64)
part_id = ((what >> 8) & 0xF).astype(int)
for kid, name in ((1, "raster_fwd_mixed"), (2, "raster_bwd_mixed")):
    m = np.nonzero(kernel == kid)[0]
    if m.size == 0:
        continue
    lo = t0[m].min()
    order = m[np.argsort(-(t1[m] - t0[m]))[:12]]
    out[name]["longest_jobs"] = [{"us": float((t1[i] - t0[i]) / 100.0), "start_us": float((t0[i] - lo) / 100.0),
                                  "tile": int(tile_id[i]), "strip": int(strip[i]), "part": int(part_id[i]),
                                  "parts": int(parts[i]), "list_len": int(lens[tile_id[i]]) if tile_id[i] < lens.size else -1,
                                  "staging_us": float(stage_us[i])} for i in order]
text = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text)
