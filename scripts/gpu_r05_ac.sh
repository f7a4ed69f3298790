#!/bin/bash
# round 5, visit AC: prefix 512 / threshold 768 behind the walk-based switch -- the whole GPU suite, then all layouts
out=gpurun_out/r05_ac
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 -x > $out/pytest.log 2>&1
tail -4 $out/pytest.log
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10 needles:0.3:10 uniform; do
  run dflt $lay
done
run always_c54 clustered:0.5:0.4 FG_HEAVY_TILES=always
run always_uniform uniform FG_HEAVY_TILES=always
