#!/bin/bash
# Binning tests (both long-segment modes) + the clustered scenes' step / stage times.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r04_binning.sh <tag> [lib.so]'
tag=${1:-r04_binning}
out=gpurun_out/$tag
mkdir -p $out
[ -n "$2" ] && export FG_RASTER_LIB=$PWD/$2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "binning or segments or job_lists or takes_the_job or sort_pairs or isect or redoes" 2>&1 | tail -15 > $out/pytest.log
cat $out/pytest.log
for sc in "0.5 0.4" "0.8 0.2" "0.0 1.0"; do
  for m in never auto; do
    [ "$m" = never ] && [ "$sc" != "0.5 0.4" ] && continue
    FG_LONG_SEGMENTS=$m timeout 300 python scripts/clustered_check.py $sc >> $out/clustered.jsonl 2>> $out/clustered.err
  done
done
cut -c1-700 $out/clustered.jsonl; tail -3 $out/clustered.err
