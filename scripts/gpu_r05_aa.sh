#!/bin/bash
# round 5, visit AA: heavy tiles by reported long walks (+ the finer splits of uneven shapes) -- parity subset, all layouts
out=gpurun_out/r05_aa
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "heavy_tiles or wide_jobs or clustered or long_segments or graph or segmented or mixed_launch or one_call or learned" > $out/pytest.log 2>&1
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
  run never $lay FG_HEAVY_TILES=never
  run always $lay FG_HEAVY_TILES=always
done
