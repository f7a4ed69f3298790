#!/bin/bash
# round 6, visit N: job timelines of clustered 0.8 / 0.2 and 0.5 / 0.4 under cost bands and under interleaved blocks
out=gpurun_out/r06_n
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1
for L in 0.8:0.2 0.5:0.4; do
  for mode in 1 3; do
    FG_RASTER_BALANCE=$mode timeout 300 python scripts/raster_timeline.py 1000000 $out/tl_${L}_$mode.json $L > /dev/null 2> $out/tl_${L}_$mode.err
    python3 - <<PY
import json
d=json.load(open("$out/tl_${L}_$mode.json"))
print("layout $L mode $mode lists", d["lists"])
for k in ("raster_fwd_mixed","raster_bwd_mixed"):
    x=d[k]; print("  ", k, "span", round(x["span_us"],1), "jobs", x["jobs"], "resident", round(x["mean_resident_waves_per_simd"],2), "xcd finish", {a:round(b) for a,b in x["per_xcd_finish_us"].items()})
    for kk,vv in x["kinds"].items(): print("       ", kk, {a:(round(b,2) if isinstance(b,float) else b) for a,b in vv.items() if a in ("jobs","mean_us","p95_us","max_us","sum_ms","prologue_share")})
    print("     longest", [(round(j["us"]), round(j["start_us"]), j["list_len"], j["strip"], j["parts"]) for j in x["longest_jobs"][:6]])
PY
  done
done
