#!/bin/bash
# Launch policy at the reference's half-resolution phase (960x540 / 300k) and at 1280x720 / 500k: list shares per tile,
# forward tails.  Usage: gpurun -- 'bash scripts/gpu_mid_res_policy.sh <tag>'
tag=${1:-midres}
out=gpurun_out/$tag
mkdir -p $out
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1
for round in 1 2; do
for cfg in "300000 960 540" "500000 1280 720"; do
  set -- $cfg
  for e in "" "FG_RASTER_SEG_PARTS=8" "FG_RASTER_SEG_PARTS=9" "FG_RASTER_SEG_PARTS=10" "FG_RASTER_SEG_PARTS=11" "FG_RASTER_TAIL_FWD=128" "FG_RASTER_TAIL_FWD=200" "FG_RASTER_TAIL_FWD=0,100000"; do
    env $e timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --n-gauss $1 --width $2 --height $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('N=$1 $2x$3 [$e]', 'sum %.4f' % sum(s.values()), 'bwd %.4f fwd %.4f' % (s.get('fg_raster_bwd',0), s.get('fg_raster_fwd',0)))" | tee -a $out/policy.txt
  done
done
done
