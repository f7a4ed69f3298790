#!/bin/bash
# round 5, visit AB: with heavy tiles on only where strips walk far, prefix length x threshold on the layout that has them
out=gpurun_out/r05_ab
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
run p1536_2560 clustered:0.8:0.2
run p1536_1792 clustered:0.8:0.2 FG_HEAVY_TILE_LEN=1792
for P in 1024 768 512 256; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_WIDE_PREFIX=$P" -j16 > ../../$out/make_$P.log 2>&1
  cd ../..
  for thr in $((P+256)) $((P+512)) $((P+1024)) 2560; do
    run p${P}_$thr clustered:0.8:0.2 FG_HEAVY_TILE_LEN=$thr
  done
  run p${P}_$((P+512))_c54 clustered:0.5:0.4 FG_HEAVY_TILE_LEN=$((P+512))
done
