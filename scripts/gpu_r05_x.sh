#!/bin/bash
# round 5, visit X: the forward's content-split thresholds on the layouts (the timeline of 0.5 / 0.4: the launch ends on
# whole-tile / two-strip jobs over lists of 950-1250 entries that started in its first microsecond)
out=gpurun_out/r05_x
mkdir -p $out
export TMPDIR=/tmp
make -C freegaussian_amd/csrc > /dev/null 2>&1
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10 uniform; do
  run dflt $lay
  for sp in 16,12 14,10 12,8 10,7 8,6 6,4; do run s${sp/,/_} $lay FG_RASTER_SPLIT_FWD=$sp; done
  for pr in 150,250 200,300 120,200; do run p${pr/,/_} $lay FG_RASTER_PRIO_FWD=$pr; done
  run s12_8_p150_250 $lay FG_RASTER_SPLIT_FWD=12,8 FG_RASTER_PRIO_FWD=150,250
done
