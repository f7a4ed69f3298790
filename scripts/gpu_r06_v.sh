#!/bin/bash
# round 6, visit V: job timelines on HEAD's policy: trained scene (two views), half of the Gaussians in a ball of 0.4
out=gpurun_out/r06_v
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1
for spec in "trained:data/trained_scene_r06.npz 3" "trained:data/trained_scene_r06.npz 6" "0.5:0.4 4"; do
  set -- $spec
  tag=$(echo $1 | sed 's/[:.\/]/_/g' | cut -c1-12)_v$2
  n=1000000
  FG_TL_VIEW=$2 FG_TL_WARM=12 timeout 300 python scripts/raster_timeline.py $n $out/tl_$tag.json $1 > /dev/null 2> $out/tl_$tag.err
  python3 - <<PY
import json
d=json.load(open("$out/tl_$tag.json"))
print("$spec lists", d["lists"])
for k in ("raster_fwd_mixed","raster_bwd_mixed"):
    x=d[k]; print("  ", k, "span", round(x["span_us"],1), "jobs", x["jobs"], "resident", round(x["mean_resident_waves_per_simd"],2), "xcd finish", {a:round(b) for a,b in x["per_xcd_finish_us"].items()})
    for kk,vv in x["kinds"].items(): print("       ", kk, {a:(round(b,2) if isinstance(b,float) else b) for a,b in vv.items() if a in ("jobs","mean_us","p95_us","max_us","sum_ms","prologue_share","staging_share")})
    print("     occupancy", [(s["t_us"], s["waves_per_simd"], s["simds_idle"]) for s in x["slices"][::2]])
    print("     longest", [(round(j["us"]), round(j["start_us"]), j["list_len"], j["strip"], j["parts"]) for j in x["longest_jobs"][:8]])
PY
done
