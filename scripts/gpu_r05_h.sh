#!/bin/bash
# round 5, visit H: the band cost cap and the heavy-tile threshold were tuned on lists WITH the dead entries the footprint masks
# now drop (-12 % on round splats): re-sweep both on the clustered layouts
out=gpurun_out/r05_h
mkdir -p $out
export TMPDIR=/tmp
cd freegaussian_amd/csrc
for cap in 10 12 14 18; do
  touch jobs_build.h
  make HIPCC="/opt/rocm/bin/hipcc -DFG_BAND_COST_CAP4=$cap" -j16 > ../../$out/make_$cap.log 2>&1
  cd ../..
  for heavy in 3072 2560 2048; do
    for lay in clustered:0.8:0.2 clustered:0.5:0.4 clustered:0.5:0.4+needles:0.3:10; do
      f=$out/b_${cap}_${heavy}_${lay//[:.+]/_}
      FG_HEAVY_TILE_LEN=$heavy timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
      python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('cap4=$cap heavy=$heavy $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
    done
  done
  cd freegaussian_amd/csrc
done
