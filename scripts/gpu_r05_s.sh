#!/bin/bash
# round 5, visit S: MEDIUM tiles -- lists between wide_tiles and heavy_tiles entries by 4-wavefront workgroups beside the main
# launch -- parity, then layouts x threshold
out=gpurun_out/r05_s
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "heavy_tiles or wide_jobs or clustered or long_segments or graph or segmented or mixed_launch" > $out/pytest_medium.log 2>&1
tail -4 $out/pytest_medium.log
FG_WIDE_TILE_LEN=512 timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "heavy_tiles or clustered or segmented or mixed_launch or randomised_parity_big" > $out/pytest_medium512.log 2>&1
tail -4 $out/pytest_medium512.log
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10 needles:0.3:10 uniform; do
  run never $lay FG_HEAVY_TILES=never
  for m in 0 2048 1536 1024 768; do run m$m $lay FG_WIDE_TILE_LEN=$m; done
done
R=$GRAFT_REPO_ROOT
(cd /tmp; for lay in clustered:0.5:0.4 clustered:0.8:0.2 uniform; do
    tag=${lay//[:.+]/_}
    FG_WIDE_TILE_LEN=1024 rocprofv3 --kernel-trace -d $R/$out/prof_$tag -o p -- python3 $R/bench.py --layout $lay --steps 32 --warmup 8 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag.json 2> $R/$out/prof_$tag.err
    echo "== $tag (medium from 1024)"; python3 $R/scripts/rocprof_top.py $R/$out/prof_$tag/p_results.db 4 raster; rm -rf $R/$out/prof_$tag
  done)
