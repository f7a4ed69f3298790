#!/bin/bash
# The GPU suite under the knobs that select round-6 code paths' alternatives.
out=gpurun_out/r06_knobs
mkdir -p $out
export TMPDIR=/tmp
run() {
  name=$1; shift
  env "$@" timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 > $out/$name.txt
  echo "$name ($*): $(tail -1 $out/$name.txt)"
}
run interleave_off FG_UNEVEN_INTERLEAVE=0
run r05_uneven FG_UNEVEN_INTERLEAVE=0 FG_UNEVEN_SPLIT_FWD=12,8 FG_UNEVEN_SPLIT2_BWD=12 FG_LONG_MANY=16
run masks_by_ratio FG_MASK_KEEP_MAX=0.8
run masks_always FG_EXACT_TILES=always
run head_slices_1 FG_DP_HEAD_SLICES=1
run no_pool FG_WORKSPACE_POOL=0
run stagewise FG_STEP_CALLS=0
