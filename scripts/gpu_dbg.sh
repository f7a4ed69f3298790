#!/bin/bash
out=gpurun_out/${1:-dbg}; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x -k "graphed_model" 2>&1 | tail -12
timeout 300 python scripts/model_host_profile.py 300000 960 540 2>&1 | grep -v "^$" | head -45
echo "=== seg_parts=1"
FG_RASTER_SEG_PARTS=1 timeout 300 python scripts/model_step_bench.py 300000 30 960 540 2>/dev/null | tr -d '\n ' | cut -c1-300; echo
