#!/bin/bash
# round 5, visit G: why did the 0.8 / 0.2 layout get slower?  masks on / off; cell shapes
out=gpurun_out/r05_g
mkdir -p $out
export TMPDIR=/tmp
for exact in 0 1; do
  f=$out/bench_c82_exact$exact
  FG_EXACT_TILES=$exact timeout 300 python bench.py --layout clustered:0.8:0.2 --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); print('exact=$exact', 'median', round(d['host_step_ms']['median'],4), 'mean', round(d['ms_per_step'],4), d['stage_ms'], 'I_raster', d['config']['I_raster'], 'heavy', d['config'].get('heavy_tile_steps'), 'long', d['config'].get('long_segment_calls'))" || tail -3 $f.err
done
for lay in uniform needles:0.3:10; do timeout 300 python scripts/cell_shape_estimate.py $lay 2>&1 | tail -3; done
