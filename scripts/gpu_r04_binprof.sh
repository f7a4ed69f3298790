#!/bin/bash
# rocprofv3 kernel table of the binning kernels on the clustered scenes.
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_r04_binprof.sh <tag> "0.5 0.4" "0.8 0.2"'
tag=${1:-r04_binprof}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
[ -n "$FG_LIB_VARIANT" ] && export FG_RASTER_LIB=$PWD/$FG_LIB_VARIANT
for sc in "$@"; do
  name=${sc// /_}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$name -o stats -- python3 scripts/clustered_check.py $sc > $out/run_$name.json 2> $out/run_$name.err
  f=$(find $out/prof_$name -name "*kernel_stats.csv" | head -1)
  cp $f $out/kernel_stats_$name.csv
  find $out/prof_$name -name "*kernel_trace*" -delete
  echo "== $sc"; cut -c1-400 $out/run_$name.json
  python3 - "$out/kernel_stats_$name.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "sb_" in n or "raster_" in n or "preprocess" in n:
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} min {float(r["MinNs"])/1e3:8.1f} max {float(r["MaxNs"])/1e3:8.1f}')
PY
done
