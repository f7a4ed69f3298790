#!/bin/bash
# Classic vs mixed forward launch at other sizes.  Usage: gpurun -- 'bash scripts/gpu_sizes.sh'
run() {
  env $1 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-graph --n-gauss $2 --width $3 --height $4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$1'.ljust(46), 'N=$2 $3x$4', 'fwd %.4f bwd %.4f step %.4f I=%d' % (s['fg_raster_fwd'], s['fg_raster_bwd'], d['ms_per_step'], d['config']['I']))"
}
for cfg in "4000000 3840 2160" "2000000 2560 1440" "1000000 1600 900" "3000000 1920 1080"; do
  run "FG_RASTER_TAIL_FWD=0 FG_RASTER_TAIL_BWD=0" $cfg
  run FG_X=1 $cfg
done
