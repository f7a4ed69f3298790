#!/bin/bash
# round 6, visit C: the reference-schedule run again with the evaluation steps around the deformation net's start; job
# timelines of the raster launches on the trained scene
out=gpurun_out/r06_c
mkdir -p $out
export TMPDIR=/tmp
T=data/trained_scene_r06.npz
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1
for v in 5 0; do
  FG_TL_VIEW=$v timeout 300 python scripts/raster_timeline.py 0 $out/timeline_trained_v$v.json trained:$T > /dev/null 2> $out/timeline_v$v.err
  python3 - <<PY
import json
d=json.load(open("$out/timeline_trained_v$v.json"))
print("view $v lists", d["lists"])
for k in ("raster_fwd_mixed","raster_bwd_mixed"):
    x=d[k]; print(k, "span", round(x["span_us"],1), "jobs", x["jobs"], "resident", round(x["mean_resident_waves_per_simd"],2), "xcd finish", {a:round(b) for a,b in x["per_xcd_finish_us"].items()})
    for kk,vv in x["kinds"].items(): print("   ", kk, {a:(round(b,2) if isinstance(b,float) else b) for a,b in vv.items() if a!="marks_mean_us"})
    print("    occupancy", [(s["t_us"], s["waves_per_simd"], s["simds_idle"]) for s in x["slices"][::2]])
    print("    longest", [(round(j["us"]), round(j["start_us"]), j["list_len"], j["strip"], j["parts"]) for j in x["longest_jobs"][:8]])
PY
done
timeout 900 python scripts/train_e2e.py --steps 7000 --eval-at 500,1000,2000,2900,3000,3100,3500,4000,5000,6000,7000 --out $out/e2e > $out/train.log 2>&1
grep "held-out" $out/train.log
rm -f $out/e2e/*.ckpt $out/e2e/trained_scene.npz
