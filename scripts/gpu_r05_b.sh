#!/bin/bash
# round 5, visit B: footprint masks -- the binning tests, the suite, uniform + needle layouts with masks on / off
out=gpurun_out/r05_b
mkdir -p $out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "footprint or supertile or job_lists or one_call or long_segments or learned or compact or heavy" 2>&1 | tail -25 > $out/pytest_masks.log
tail -12 $out/pytest_masks.log
FG_PARITY_REPORT=$out/parity_margins.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -30 > $out/pytest.log
tail -8 $out/pytest.log
for exact in 1 0; do
for lay in uniform needles:0.3:10 clustered:0.5:0.4+needles:0.3:10 clustered:0.5:0.4; do
  f=$out/bench_${lay//[:.+]/_}_exact$exact
  FG_EXACT_TILES=$exact timeout 300 python bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.load(open('$f.json')); print('exact=$exact', '$lay', round(d['value'],1), 'Mpix/s', round(d['ms_per_step'],4), 'ms', d['stage_ms'], 'I_raster', d['config']['I_raster'], 'longest', d['config'].get('longest_tile_list'))" || tail -3 $f.err
done; done
