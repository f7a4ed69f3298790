#!/bin/bash
# round 5, visit Y: finer forward content thresholds for shapes that showed an uneven scene (host policy) -- all layouts, and
# the backward's thresholds on the clustered ones
out=gpurun_out/r05_y
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10 needles:0.3:10 uniform; do
  run off $lay FG_UNEVEN_SPLIT_FWD=0
  run dflt $lay
  run u10_7 $lay FG_UNEVEN_SPLIT_FWD=10,7
done
for lay in clustered:0.5:0.4 clustered:0.8:0.2; do
  for sb in 20,12 20,10 20,8 16,12; do run bwd${sb/,/_} $lay FG_RASTER_SPLIT_BWD=$sb; done
  for sg in 6 8; do run parts$sg $lay FG_RASTER_SEG_PARTS=$sg; done
done
