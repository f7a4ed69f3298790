#!/bin/bash
# The knobs that select code paths changed in round 3, each under the whole GPU suite.  Usage: gpurun -- 'bash scripts/gpu_knob_subset.sh'
for env in "FG_BINNING=depthfirst" "FG_SH_JAC=0" "FG_SPECULATIVE_BINNING=0" "FG_DIRECT_COUNT=0 FG_FILL_IN_FORWARD=0" "FG_OVERLAP_PACK=1" \
           "FG_TIGHT_RECTS=0" "FG_LONG_SEGMENTS=always" "FG_LONG_SEGMENTS=never" "FG_RASTER_SEG_PARTS=1" "FG_RASTER_PPT_FWD=1 FG_RASTER_PPT_BWD=1"; do
  res=$(env $env timeout 900 python -m pytest tests -m gpu -q -x -k "not two_ranks and not lockstep and not world_size" 2>&1 | tail -1)
  echo "$env | $res"
done
