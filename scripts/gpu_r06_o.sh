#!/bin/bash
# round 6, visit O: interleaved blocks x the backward's share threshold x the forward's content thresholds
out=gpurun_out/r06_o
mkdir -p $out
export TMPDIR=/tmp
for lay in clustered:0.8:0.2 clustered:0.5:0.4 trained:data/trained_scene_r06.npz needles:0.3:10; do
  for sb in "20,4" "20,3"; do
  for sf in "20,16" "12,8" "8,5" "6,4"; do
    FG_RASTER_SPLIT_BWD=$sb FG_RASTER_SPLIT_FWD=$sf FG_RASTER_BALANCE=3 timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay bwd=$sb fwd=$sf', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k}, d['config']['seg_ckpt_mb'])"
  done
  done
done
