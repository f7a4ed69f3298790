#!/bin/bash
# Run the GPU parity suite under the non-default runtime knobs (every template instantiation and
# optional path must give the same results).  Usage: gpurun -- 'bash scripts/gpu_knob_matrix.sh'
for env in "FG_RASTER_PPT_FWD=1 FG_RASTER_PPT_BWD=1" "FG_RASTER_PPT_FWD=4 FG_RASTER_PPT_BWD=2" "FG_RASTER_PPT_FWD=2 FG_RASTER_PPT_BWD=4" \
           "FG_RASTER_TAIL_FWD=3,4 FG_RASTER_TAIL_BWD=5,2 FG_RASTER_SPLIT_FWD=3,2 FG_RASTER_SPLIT_BWD=2,1" "FG_RASTER_TAIL_FWD=100000 FG_RASTER_TAIL_BWD=100000" \
           "FG_TILE_ORDER=rows" "FG_SPECULATIVE_BINNING=0" "FG_OVERLAP_PACK=1" \
           "FG_TIGHT_RECTS=0" "FG_RASTER_LIVE=0" "FG_RASTER_SEG_PARTS=1" "FG_RASTER_SEG_PARTS=5 FG_RASTER_SEG_TAIL=0" \
           "FG_RASTER_LIVE=0 FG_RASTER_SEG_PARTS=1 FG_RASTER_TAIL_BWD=7,9 FG_RASTER_SPLIT_BWD=2,1" "FG_BINNING=depthfirst" "FG_SH_JAC=0" "FG_LONG_SEGMENTS=always" "FG_LONG_SEGMENTS=never" "FG_JOBS_IN_FILL=0" \
           "FG_DIRECT_COUNT=0 FG_FILL_IN_FORWARD=0" "FG_RASTER_BANDS=8" "FG_RASTER_BANDS=4 FG_RASTER_SEG_PARTS=7" "FG_RASTER_BANDS=2 FG_RASTER_SEG_GRADE=6,150" \
           "FG_RASTER_SEG_PARTS=16 FG_RASTER_SEG_TAIL=0"; do
  res=$(env $env timeout 900 python -m pytest tests -m gpu -q -x -k "not two_ranks and not lockstep and not world_size" 2>&1 | tail -1)
  echo "$env | $res"
done
