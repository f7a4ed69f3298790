#!/bin/bash
# round 5, visit AI: the switches of the round still work -- the raster tests under each of them
out=gpurun_out/r05_ai
mkdir -p $out
export TMPDIR=/tmp
K="heavy_tiles or wide_jobs or clustered or segmented or compact or one_call or learned or graph or mixed_launch or launch_policies or randomised_parity_big"
for env in "FG_RASTER_HEAVY_WIDE=0" "FG_RASTER_SEG_FINE=0" "FG_HEAVY_TILES=always" "FG_HEAVY_TILES=never" "FG_UNEVEN_SPLIT_FWD=0" "FG_STEP_CALLS=0" "FG_RASTER_SEG_FINE=256" "FG_HEAVY_TILE_LEN=1792 FG_HEAVY_TILES=always"; do
  tag=${env// /_}
  env $env timeout 900 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "$K" > $out/pytest_$tag.log 2>&1
  echo "$env: $(tail -1 $out/pytest_$tag.log)"
  grep -E "^FAILED" $out/pytest_$tag.log | head -5
done
