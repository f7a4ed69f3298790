#!/bin/bash
# A/B of library builds (FG_RASTER_LIB) on the bench scene: step time, stage times, kernel averages of the bin kernels.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_variant_ab.sh <tag> <lib.so> [<lib.so> ...]'   (paths under freegaussian_amd/)
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1  # (the first process on a fresh box runs slow)
for round in 1 2; do
for lib in "$@"; do
  FG_RASTER_LIB=$PWD/freegaussian_amd/$lib timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph 2>$out/err_$lib.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib'.ljust(26), round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'])" | tee -a $out/ab.txt
done
done
for lib in "$@"; do
  FG_RASTER_LIB=$PWD/freegaussian_amd/$lib timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$lib -o stats -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2> $out/prof_$lib.err
  f=$(find $out/prof_$lib -name "*kernel_stats*" | head -1)
  echo "== $lib" | tee -a $out/ab.txt
  python3 - <<PY | tee -a $out/ab.txt
import csv
for r in list(csv.DictReader(open("$f")))[:14]:
    if "sb_" in r["Name"] or "build_jobs" in r["Name"]:
        print(f'  {r["Name"][:70]:70s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
  find $out/prof_$lib -name "*kernel_trace*" -delete
done
