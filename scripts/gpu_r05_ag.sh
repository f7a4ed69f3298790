#!/bin/bash
# round 5, visit AG: the bands' cost cap again (quarters of the mean list a tile's cost is capped at), after the wide jobs, the
# finer splits and the checkpoint grid
out=gpurun_out/r05_ag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'))" || tail -2 $f.err
}
for cap in 12 4 6 8 10 16; do
  cd freegaussian_amd/csrc
  touch jobs_build.h
  make HIPCC="/opt/rocm/bin/hipcc -DFG_BAND_COST_CAP4=$cap" -j16 > ../../$out/make_$cap.log 2>&1
  cd ../..
  for lay in clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10; do run cap$cap $lay; done
done
