#!/bin/bash
# Sweep of the mixed launch's tail size (tiles per XCD split into single-strip jobs).
# Usage: gpurun -- 'bash scripts/gpu_tail_sweep.sh FWD "0 64 128 256 384 512"'
which=$1
for t in $2; do
  env FG_RASTER_TAIL_$which=$t timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('tail_$which=$t', 'fwd %.4f bwd %.4f step %.4f' % (s['fg_raster_fwd'], s['fg_raster_bwd'], d['ms_per_step']))"
done
