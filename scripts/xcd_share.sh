#!/bin/bash
# Per-XCD share of the raster kernels' work: one bench run per XCD with FG_DEBUG_ONLY_XCD=x (only
# that XCD's workgroups do their tiles), next to the full launch.
# Usage: gpurun -- 'bash scripts/xcd_share.sh [extra env]'
for x in all 0 1 2 3 4 5 6 7; do
  if [ $x = all ]; then e=""; else e="FG_DEBUG_ONLY_XCD=$x"; fi
  env $e "$@" timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('xcd $x', 'fwd %.4f bwd %.4f' % (s['fg_raster_fwd'], s['fg_raster_bwd']))"
done
