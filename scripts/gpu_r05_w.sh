#!/bin/bash
# round 5, visit W: job timeline of the clustered layouts on HEAD (wide jobs): what the forward's launches end on
out=gpurun_out/r05_w
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1; tail -2 $out/make_timeline.log
for sc in "0.5:0.4" "0.8:0.2"; do
  name=${sc//:/_}
  timeout 300 python scripts/raster_timeline.py 1000000 $out/timeline_$name.json "$sc" > /dev/null 2> $out/timeline_$name.err
  FG_HEAVY_TILES=never timeout 300 python scripts/raster_timeline.py 1000000 $out/timeline_never_$name.json "$sc" > /dev/null 2> $out/timeline_never_$name.err
done
python - <<PY
import json
for n in ("0.5_0.4", "never_0.5_0.4", "0.8_0.2"):
    try:
        d = json.load(open("$out/timeline_%s.json" % n))
    except Exception as e:
        print(n, "failed", e); continue
    print("==", n, d.get("lists"))
    for k in ("raster_fwd_mixed",):
        v = d.get(k)
        if not v: continue
        print(k, "span", round(v["span_us"], 1), "jobs", v["jobs"], "resident", round(v["mean_resident_waves_per_simd"], 2),
              "xcd finish", {a: round(b) for a, b in v["per_xcd_finish_us"].items()})
        print("  kinds", {a: (b["jobs"], round(b["mean_us"], 1), round(b["max_us"], 1), round(b["sum_ms"], 2)) for a, b in v["kinds"].items()})
        print("  occupancy", [s["waves_per_simd"] for s in v["slices"]])
        for j in v["longest_jobs"][:12]: print("   ", j)
PY
