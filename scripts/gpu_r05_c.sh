#!/bin/bash
# round 5, visit C: footprint masks after the count kernel's row-wise marks -- tests, uniform / needles A/B by stage
out=gpurun_out/r05_c
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -s -k "footprint or randomised or second_backward" 2>&1 | grep -E "isotropic:|needles:|passed|failed|FAILS|MISMATCH|Error|assert" | cut -c1-400 > $out/pytest_masks.log
tail -40 $out/pytest_masks.log
for rep in 1 2; do
for exact in 1 0; do
for lay in uniform needles:0.3:10; do
  f=$out/bench_${lay//[:.+]/_}_exact${exact}_$rep
  FG_EXACT_TILES=$exact timeout 300 python bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.load(open('$f.json')); print('exact=$exact', '$lay', 'median', round(d['host_step_ms']['median'],4), 'mean', round(d['ms_per_step'],4), 'ms', d['stage_ms'], 'I_raster', d['config']['I_raster'])" || tail -3 $f.err
done; done; done
