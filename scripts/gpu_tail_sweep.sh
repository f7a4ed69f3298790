#!/bin/bash
# Sweep of the mixed launch's tail sizes (tiles per XCD: "t4" quarters or "t4,t2" quarters,halves).
# Usage: gpurun -- 'bash scripts/gpu_tail_sweep.sh FWD "0 256 510 255,510" [steps]'
which=$1
steps=${3:-50}
for t in $2; do
  env FG_RASTER_TAIL_$which=$t timeout 200 python bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('tail_$which=$t', 'fwd %.4f bwd %.4f step %.4f' % (s['fg_raster_fwd'], s['fg_raster_bwd'], d['ms_per_step']))"
done
