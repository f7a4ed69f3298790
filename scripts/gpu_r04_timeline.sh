#!/bin/bash
# Round-4 first visit: job timelines of the two mixed raster launches on HEAD (uniform bench scene and the two
# clustered scenes of scripts/clustered_check.py) + the clustered step times, stage by stage.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r04_timeline.sh <tag>'
tag=${1:-r04_timeline}
out=gpurun_out/$tag
mkdir -p $out
make -C freegaussian_amd/csrc timeline > $out/make.log 2>&1
for sc in "" "0.5:0.4" "0.8:0.2"; do
  name=${sc:-uniform}; name=${name//:/_}
  timeout 300 python scripts/raster_timeline.py 1000000 $out/timeline_$name.json "$sc" > /dev/null 2> $out/timeline_$name.err
done
for sc in "0.5 0.4" "0.8 0.2"; do
  for b in supertile depthfirst; do
    FG_BINNING=$b timeout 300 python scripts/clustered_check.py $sc >> $out/clustered.jsonl 2>> $out/clustered.err
  done
  timeout 300 python scripts/clustered_check.py $sc >> $out/clustered.jsonl 2>> $out/clustered.err
done
cat $out/clustered.jsonl | cut -c1-900
python - <<PY
import json
for n in ("uniform", "0.5_0.4", "0.8_0.2"):
    try:
        d = json.load(open("$out/timeline_%s.json" % n))
    except Exception as e:
        print(n, "failed", e); continue
    print("==", n, d.get("lists"))
    for k in ("raster_fwd_mixed", "raster_bwd_mixed"):
        v = d.get(k)
        if not v: continue
        print(k, "span", round(v["span_us"], 1), "jobs", v["jobs"], "resident", round(v["mean_resident_waves_per_simd"], 2),
              "xcd finish", {a: round(b) for a, b in v["per_xcd_finish_us"].items()})
        print("  kinds", {a: (b["jobs"], round(b["mean_us"], 1), round(b["max_us"], 1), round(b["sum_ms"], 2)) for a, b in v["kinds"].items()})
        print("  occupancy", [s["waves_per_simd"] for s in v["slices"]])
        print("  idle", [s["simds_idle"] for s in v["slices"]])
        for j in v["longest_jobs"][:6]: print("   ", j)
PY
