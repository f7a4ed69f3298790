#!/bin/bash
# round 5, visit AJ: the length of a walk the forward reports (heavy tiles stay on for a shape while strips walk that far)
out=gpurun_out/r05_aj
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'heavy_steps', d['config'].get('heavy_tile_steps'), 'longest', d['config'].get('longest_tile_list'))" || tail -2 $f.err
}
for R in 1024 1536 2048 2560; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_WALK_REPORT=$R" -j16 > ../../$out/make_$R.log 2>&1
  cd ../..
  for lay in needles:0.3:10 clustered:0.5:0.4 clustered:0.5:0.4+needles:0.3:10 clustered:0.65:0.3 clustered:0.8:0.2; do run r$R $lay FG_HEAVY_FLAG_LEN=$((R < 2560 ? R : 2560)); done
done
