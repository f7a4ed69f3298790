#!/bin/bash
out=gpurun_out/${1:-dbg}; mkdir -p $out
for b in 0 1; do
  echo "== banded=$b 65536 test"
  FG_BANDED_BINNING=$b AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 300 python -m pytest tests -m gpu -q -x -k "65536" 2>&1 | grep -v "^  File\|pluggy\|^$" | tail -12
done
echo "== banded tests"
timeout 600 python -m pytest tests -m gpu -q -k "banded" 2>&1 | tail -15
echo "== graphed model"
timeout 600 python -m pytest tests -m gpu -q -x -k "graphed_model" 2>&1 | tail -30
for sz in "100000 30 480 270" "300000 30 960 540"; do timeout 300 python scripts/model_step_bench.py $sz 2>&1 | tail -22 | head -8; done
