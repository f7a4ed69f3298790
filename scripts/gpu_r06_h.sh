#!/bin/bash
# round 6, visit H: per-XCD finish times, cost-balanced bands vs equal spans, on two layouts of the regret table
out=gpurun_out/r06_h
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1
L1='{"n": 1000000, "cam_radius": 2.5, "scale_mean": 0.01, "scale_max": 0.15, "opac_std": 1.5, "opac_shift": -1.0, "layout": "clustered:0.6:0.4+needles:0.1:20"}'
L2='{"n": 300000, "cam_radius": 2.5, "scale_mean": 0.005, "scale_max": 0.15, "opac_std": 1.5, "opac_shift": -1.0, "layout": "clustered:0.8:0.4+needles:0.1:10"}'
i=0
for L in "$L1" "$L2"; do
  i=$((i+1))
  for mode in auto equal; do
    if [ $mode = equal ]; then export FG_RASTER_BALANCE=2; else unset FG_RASTER_BALANCE; fi
    timeout 300 python scripts/raster_timeline.py 0 $out/tl_${i}_$mode.json "$L" > /dev/null 2> $out/tl_${i}_$mode.err
    python3 - <<PY
import json
d=json.load(open("$out/tl_${i}_$mode.json"))
print("layout $i $mode lists", d["lists"])
for k in ("raster_fwd_mixed","raster_bwd_mixed"):
    x=d[k]; print("  ", k, "span", round(x["span_us"],1), "resident", round(x["mean_resident_waves_per_simd"],2), "xcd finish", {a:round(b) for a,b in x["per_xcd_finish_us"].items()})
    print("     longest", [(round(j["us"]), round(j["start_us"]), j["list_len"], j["strip"], j["parts"]) for j in x["longest_jobs"][:6]])
PY
  done
done
