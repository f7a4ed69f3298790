#!/bin/bash
# round 6, visit W: are unsplit backward tiles of the trained scene short of checkpoint slots or of job slots?
out=gpurun_out/r06_w
mkdir -p $out
export TMPDIR=/tmp
for lay in trained:data/trained_scene_r06.npz clustered:0.5:0.4; do
for e in "X=1" "FG_COMPACT_SLOTS=0" "FG_RASTER_SEG_PARTS=6" "FG_RASTER_SEG_TAIL=0" "FG_UNEVEN_SPLIT2_BWD=3" "FG_UNEVEN_SPLIT2_BWD=6"; do
  env $e timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
  python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay $e', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k}, d['config']['seg_ckpt_mb'])"
done
done
