#!/bin/bash
# focused debugging visit: gpurun -- 'bash scripts/gpu_debug.sh <tag> <pytest -k expression>'
out=gpurun_out/${1:-dbg}; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -x -q -k "$2" 2>&1 | tail -80 > $out/pytest.log
cat $out/pytest.log | cut -c1-220
