#!/bin/bash
# round 5, visit V: medium tiles, workgroups of the launch
out=gpurun_out/r05_v
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
for G in 8192 32768; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_MEDIUM_GRID=$G" -j16 > ../../$out/make_$G.log 2>&1
  cd ../..
  for lay in clustered:0.5:0.4 uniform needles:0.3:10; do
    for m in 1536 1024; do run g${G}_m$m $lay FG_WIDE_TILE_LEN=$m; done
  done
done
