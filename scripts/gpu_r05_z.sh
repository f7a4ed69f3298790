#!/bin/bash
# round 5, visit Z: with the finer splits on uneven shapes -- the wide jobs' prefix length again, the backward's threshold in
out=gpurun_out/r05_z
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
}
LAYS="clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10"
for lay in $LAYS; do run p1536 $lay; run p1536_never $lay FG_HEAVY_TILES=never; done
for P in 1024 768 512; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_WIDE_PREFIX=$P" -j16 > ../../$out/make_$P.log 2>&1
  cd ../..
  for lay in $LAYS; do
    run p${P}_2560 $lay
    run p${P}_min $lay FG_HEAVY_TILE_LEN=$((P+512))
  done
done
