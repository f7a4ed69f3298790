#!/bin/bash
# Backward list shares per tile (FG_RASTER_SEG_PARTS) at the reference's low-resolution phases: is the default (6 below
# 5000 tiles) still right?  Usage: gpurun -- 'bash scripts/gpu_low_res_parts.sh <tag>'
tag=${1:-lowres_parts}
out=gpurun_out/$tag
mkdir -p $out
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1
for cfg in "100000 480 270" "300000 960 540" "500000 1280 720"; do
  set -- $cfg
  for parts in default 4 8 10 12 16; do
    env=""; [ $parts != default ] && env="FG_RASTER_SEG_PARTS=$parts"
    env $env timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --n-gauss $1 --width $2 --height $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('N=$1 $2x$3 parts=$parts', '%.4f ms' % d['ms_per_step'], 'bwd %.4f fwd %.4f' % (s.get('fg_raster_bwd',0), s.get('fg_raster_fwd',0)))" | tee -a $out/parts.txt
  done
done
