#!/bin/bash
# The knobs that select code paths added in round 4, each under the whole GPU suite.  Usage: gpurun -- 'bash scripts/gpu_knob_r04.sh'
out=gpurun_out/r04_knobs; mkdir -p $out
for env in "FG_STEP_CALLS=0" "FG_COMPACT_SLOTS=0" "FG_EVEN_BANDS=0 FG_RASTER_PRIO_FWD=0 FG_RASTER_PRIO_BWD=0" "FG_HEAVY_TILES=always FG_LONG_SEGMENTS=always" \
           "FG_RASTER_BALANCE=100" "FG_RASTER_BALANCE=0 FG_HEAVY_TILES=never" "FG_RASTER_BALANCE=2"; do
  res=$(env $env timeout 900 python -m pytest tests -m gpu -q -x -k "not two_ranks and not lockstep and not world_size" 2>&1 | tail -1)
  echo "$env: $res" | tee -a $out/knobs.txt
done
