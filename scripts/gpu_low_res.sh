#!/bin/bash
# The raster step at the reference's low-resolution phases (eager and graphed).  Usage: gpurun -- 'bash scripts/gpu_low_res.sh <tag>'
tag=${1:-lowres}
out=gpurun_out/$tag
mkdir -p $out
for cfg in "100000 480 270" "300000 960 540" "500000 1280 720" "1000000 1600 900"; do
  set -- $cfg
  timeout 400 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --n-gauss $1 --width $2 --height $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('N=$1 $2x$3', 'eager %.4f ms (%.0f Mpix/s)' % (d['ms_per_step'], d['value']), 'graphed %.4f' % d.get('graphed',{}).get('ms_per_step', float('nan')), 'stages sum %.4f' % sum(s.values()), {k.replace('fg_',''): v for k, v in s.items()})" | tee -a $out/low_res.txt
done
