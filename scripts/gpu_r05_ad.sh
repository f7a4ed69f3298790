#!/bin/bash
# round 5, visit AD: the backward on the clustered layouts -- how many tiles are cut into shares, and how finely
out=gpurun_out/r05_ad
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 uniform; do
  run dflt $lay
  for t in 600 800 1020; do run tail$t $lay FG_RASTER_SEG_TAIL=$t; done
  run tail1020_p3 $lay FG_RASTER_SEG_TAIL=1020 FG_RASTER_SEG_PARTS=3
  run tail1020_p4 $lay FG_RASTER_SEG_TAIL=1020 FG_RASTER_SEG_PARTS=4
  for s2 in 8 6 4; do run s2_$s2 $lay FG_UNEVEN_SPLIT2_BWD=$s2; done
  run s2_6_tail800 $lay FG_UNEVEN_SPLIT2_BWD=6 FG_RASTER_SEG_TAIL=800
done
