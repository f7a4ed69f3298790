#!/bin/bash
# round 5, visit F: evidence on HEAD -- raster work counters, job timelines (uniform + two clustered scenes), kernel tables of
# the clustered layouts, the default bench line
out=gpurun_out/r05_f
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc stats > $out/make_stats.log 2>&1
timeout 300 python scripts/raster_stats.py > $out/raster_work_counters.json 2> $out/raster_stats.err; cat $out/raster_work_counters.json | tr -d '\n' | cut -c1-900; echo
make -C freegaussian_amd/csrc timeline > $out/make_timeline.log 2>&1
for sc in "" "0.5:0.4" "0.8:0.2"; do
  name=${sc:-uniform}; name=${name//:/_}
  timeout 300 python scripts/raster_timeline.py 1000000 $out/timeline_$name.json "$sc" > /dev/null 2> $out/timeline_$name.err
done
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
        for j in v["longest_jobs"][:4]: print("   ", j)
PY
for lay in clustered:0.5:0.4 clustered:0.8:0.2; do
  tag=${lay//[:.+]/_}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o stats -- python3 bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/prof_$tag.json 2> $out/prof_$tag.err
  find $out/prof_$tag -name "*kernel_stats*" -exec cp {} $out/kernel_stats_$tag.csv \;
  rm -rf $out/prof_$tag
done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-1500
