#!/bin/bash
# Kernel timeline of the bench step: where is the GPU idle between launches?
# Usage: gpurun --timeout 900 -- 'bash scripts/gpu_gaps.sh <tag> [bench flags]'
tag=${1:-gaps}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-graph "$@" > $out/bench.json 2> $out/err.log
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python scripts/trace_gaps.py $f > $out/gaps.txt 2>&1
rm -f $f
cat $out/gaps.txt; cut -c1-400 $out/bench.json
