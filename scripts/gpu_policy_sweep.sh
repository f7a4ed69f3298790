#!/bin/bash
# One-at-a-time sweep of the raster launch policy around its defaults on the bench scene (steady-state bench).
# Usage: gpurun -- 'bash scripts/gpu_policy_sweep.sh <tag>'
tag=${1:-policy}
out=gpurun_out/$tag
mkdir -p $out
run() {
  env $1 timeout 300 python bench.py --no-clustered --steps 48 --warmup 8 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$1'.ljust(34), round(d['value'],1), round(d['ms_per_step'],4), 'bwd', s['fg_raster_bwd'], 'fwd', s['fg_raster_fwd'])" | tee -a $out/sweep.txt
}
run FG_X=0
for v in 3 4 6 7; do run FG_RASTER_SEG_PARTS=$v; done
for v in 250 550 700 1020; do run FG_RASTER_SEG_TAIL=$v; done
for v in 380 640 800; do run FG_RASTER_TAIL_FWD=$v; done
for v in "14,10" "28,22" "0,0"; do run FG_RASTER_SPLIT_BWD=$v; done
for v in "14,10" "28,22"; do run FG_RASTER_SPLIT_FWD=$v; done
run FG_X=0
