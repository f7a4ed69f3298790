#!/bin/bash
# scripts/stbin_micro.py under rocprofv3 for several library builds.  Usage: gpurun -- 'bash scripts/gpu_stbin_micro.sh <tag> <lib.so> ...'
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for lib in "$@"; do
  FG_RASTER_LIB=$PWD/freegaussian_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$lib -o stats -- python3 scripts/stbin_micro.py $SB_ARGS 2> $out/err_$lib.txt | tee -a $out/micro.txt
  f=$(find $out/prof_$lib -name "*kernel_stats*" | head -1)
  python3 - <<PY | tee -a $out/micro.txt
import csv
for r in csv.DictReader(open("$f")):
    if "sb_" in r["Name"]:
        print(f'  {r["Name"][:60]:60s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
  find $out/prof_$lib -name "*kernel_trace*" -delete
done
