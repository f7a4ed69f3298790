#!/bin/bash
# round 5, visit I: ids of the next batch prefetched in the forward; re-tuned clustered thresholds -- stage times per layout,
# forward parity tests
out=gpurun_out/r05_i
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "mixed_launch or heavy or clustered or raster_forward or cfg4 or one_call or randomised_parity_big" 2>&1 | tail -5
for rep in 1 2; do
for lay in uniform clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10; do
  f=$out/b_${lay//[:.+]/_}_$rep
  timeout 200 python bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$lay', 'median', round(d['host_step_ms']['median'],4), 'mean', round(d['ms_per_step'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'count', s.get('fg_bin_prepare'))" || tail -2 $f.err
done; done
