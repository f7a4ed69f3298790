#!/bin/bash
# round 5, visit A: the whole GPU suite (with the randomised parity cases in the gate), the step series across refinements,
# the needle layouts on the unchanged binning
out=gpurun_out/r05_a
mkdir -p $out
export TMPDIR=/tmp
FG_PARITY_REPORT=$out/parity_margins.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=15 2>&1 | tail -60 > $out/pytest.log
tail -30 $out/pytest.log
for lay in uniform clustered:0.5:0.4; do
  timeout 600 python scripts/refine_step_bench.py $lay > $out/refine_${lay//[:.]/_}.json 2> $out/refine_${lay//[:.]/_}.err
  tail -c 2500 $out/refine_${lay//[:.]/_}.json; tail -3 $out/refine_${lay//[:.]/_}.err
done
for lay in needles:0.3:10 clustered:0.5:0.4+needles:0.3:10 needles:1.0:10; do
  f=$out/bench_${lay//[:.+]/_}
  timeout 300 python bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.load(open('$f.json')); print('$lay', round(d['value'],1), 'Mpix/s', round(d['ms_per_step'],4), 'ms', d['stage_ms'], 'I_raster', d['config']['I_raster'], 'longest', d['config'].get('longest_tile_list'))" || tail -3 $f.err
done
