#!/bin/bash
# round 5, visit P: heavy tiles = prefix jobs in the main launch + wide jobs behind it -- parity, then layouts x prefix x threshold
out=gpurun_out/r05_p
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "heavy_tiles or clustered or long_segments or graph or segmented" > $out/pytest_wide.log 2>&1
tail -4 $out/pytest_wide.log
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'heavy_steps', d['config'].get('heavy_tile_steps'))" || tail -2 $f.err
}
LAYS="clustered:0.8:0.2 clustered:0.5:0.4 clustered:0.5:0.4+needles:0.3:10 needles:0.3:10"
for lay in $LAYS; do
  run never $lay FG_HEAVY_TILES=never
  run three_2560 $lay FG_RASTER_HEAVY_WIDE=0
  run p1536_2560 $lay FG_HEAVY_TILE_LEN=2560
  run p1536_1792 $lay FG_HEAVY_TILE_LEN=1792
done
R=$GRAFT_REPO_ROOT
(cd /tmp; for lay in clustered:0.8:0.2 clustered:0.5:0.4; do
    tag=${lay//[:.+]/_}
    rocprofv3 --kernel-trace -d $R/$out/prof_$tag -o p -- python3 $R/bench.py --layout $lay --steps 32 --warmup 8 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag.json 2> $R/$out/prof_$tag.err
    echo "== $tag (prefix 1536, heavy 2560)"; python3 $R/scripts/rocprof_top.py $R/$out/prof_$tag/p_results.db 3 raster; rm -rf $R/$out/prof_$tag
  done)
for P in 1024 2048; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_WIDE_PREFIX=$P" -j16 > ../../$out/make_$P.log 2>&1
  cd ../..
  for lay in $LAYS; do
    run p${P}_2560 $lay FG_HEAVY_TILE_LEN=2560
    run p${P}_min $lay FG_HEAVY_TILE_LEN=$((P+256))
  done
done
