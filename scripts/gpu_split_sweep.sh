#!/bin/bash
# Sweep of the content thresholds of the job lists ("a4,a2" in 1/65536 of the total list length) on
# the uniform bench scene and on the clustered scene.
# Usage: gpurun -- 'bash scripts/gpu_split_sweep.sh "32,16 32,12 24,12" [FWD|BWD|BOTH]'
which=${2:-BOTH}
for t in $1; do
  if [ $which = BOTH ]; then e="FG_RASTER_SPLIT_FWD=$t FG_RASTER_SPLIT_BWD=$t"; else e="FG_RASTER_SPLIT_$which=$t"; fi
  env $e timeout 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('uniform   $which=$t'.ljust(28), 'fwd %.4f bwd %.4f step %.4f' % (s['fg_raster_fwd'], s['fg_raster_bwd'], d['ms_per_step']))"
  env $e timeout 200 python scripts/clustered_check.py 0.5 0.4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('clustered $which=$t'.ljust(28), 'fwd %.4f bwd %.4f step %.4f' % (d['fwd'], d['bwd'], d['step_ms']))"
done
