#!/bin/bash
timeout 600 python -m pytest tests -m gpu -q -x -k "graphed_model or graph_replays or graphed_raster or world_size_8" 2>&1 | grep -v "^  File\|Warning\|warn" | tail -15
for sz in "100000 30 480 270" "300000 30 960 540"; do timeout 300 python scripts/model_step_bench.py $sz 2>/dev/null | tr -d '\n ' | cut -c1-330; echo; done
