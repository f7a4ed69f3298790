#!/bin/bash
# round 5, visit L: heavy tiles as WIDE jobs (fg_raster_config::heavy_wide) -- parity first (both forms), then the heavy
# threshold swept on the clustered layouts, then rocprof kernel averages of the best
out=gpurun_out/r05_l
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "heavy_tiles or clustered or long_segments" > $out/pytest_wide.log 2>&1
tail -3 $out/pytest_wide.log
FG_RASTER_HEAVY_WIDE=0 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "heavy_tiles or clustered" > $out/pytest_three.log 2>&1
tail -2 $out/pytest_three.log
run() {  # name, env..., layout
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
for lay in clustered:0.8:0.2 clustered:0.5:0.4 clustered:0.5:0.4+needles:0.3:10; do
  run three_2560 $lay FG_RASTER_HEAVY_WIDE=0 FG_HEAVY_TILE_LEN=2560
  for heavy in 2560 2048 1536 1024 768 512; do
    run wide_$heavy $lay FG_HEAVY_TILE_LEN=$heavy
  done
done
run wide_1024 uniform FG_HEAVY_TILE_LEN=1024
run wide_1024_always uniform FG_HEAVY_TILE_LEN=1024 FG_HEAVY_TILES=always
cd /tmp
for heavy in 2560 1024; do
  FG_HEAVY_TILE_LEN=$heavy rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof_$heavy -o p -- python3 $GRAFT_REPO_ROOT/bench.py --layout clustered:0.8:0.2 --steps 32 --warmup 8 --no-cpu-baseline --no-graph --no-clustered > $GRAFT_REPO_ROOT/$out/prof_$heavy.json 2> $GRAFT_REPO_ROOT/$out/prof_$heavy.err
  f=$(find $GRAFT_REPO_ROOT/$out/prof_$heavy -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $GRAFT_REPO_ROOT/$out/kernel_stats_0.8_0.2_$heavy.csv && head -14 $f | cut -d, -f1-4 | cut -c1-150
done
