#!/bin/bash
# round 5, visit K: repeated backward through the one-call node at full size (the fuzz sweep's case), capacity hysteresis
out=gpurun_out/r05_k
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "second_backward or learned or one_call or speculative or redoes" 2>&1 | tail -4
FUZZ_ONLY=15 timeout 900 python scripts/fuzz_parity.py 24 4 big 2>&1 | tail -3 | cut -c1-700
for lay in clustered:0.5:0.4 uniform; do
  timeout 600 python scripts/refine_step_bench.py $lay > $out/refine_${lay//[:.]/_}.json 2> $out/refine_${lay//[:.]/_}.err
  python3 -c "
import json; d=json.load(open('$out/refine_${lay//[:.]/_}.json')); print('$lay', 'series / steady', d['series_mean_over_steady_state'], 'without stalls', d['series_mean_over_steady_state_without_those'], d['steps_over_5x_steady'], d['counters_after_the_first_calls'], [(s['n_gauss'], s['mean_ms'], s['steady_ms'], s['first8_ms'][:2]) for s in d['segments']])"
done
