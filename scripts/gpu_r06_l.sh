#!/bin/bash
# round 6, visit L: interleaved blocks with the cost bands' larger grids (room for content splits)
out=gpurun_out/r06_l
mkdir -p $out
export TMPDIR=/tmp
for lay in uniform clustered:0.5:0.4 clustered:0.8:0.2 trained:data/trained_scene_r06.npz; do
  for m in 1 3 4; do
    FG_RASTER_BALANCE=$m timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay balance=$m', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k or 'emit' in k})"
  done
done
