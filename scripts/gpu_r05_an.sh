#!/bin/bash
# round 5, visit AN: the one-call path's workspaces from a pool of the context's own -- the gate, the refine series (default
# allocator settings), the headline
out=gpurun_out/r05_an
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu --timeout 900 > $out/pytest.log 2>&1; tail -1 $out/pytest.log; grep -E "^FAILED" $out/pytest.log | head
for lay in uniform clustered:0.5:0.4; do
  for pool in 1 0; do
    FG_WORKSPACE_POOL=$pool REFINE_DIAG=1 timeout 500 python scripts/refine_step_bench.py $lay > $out/refine_${pool}_${lay//[:.]/_}.json 2> $out/refine_${pool}_${lay//[:.]/_}.err
    python3 -c "
import json; d=json.load(open('$out/refine_${pool}_${lay//[:.]/_}.json')); print('pool=$pool', '$lay', d['series_mean_over_steady_state'], d['steps_over_5x_steady'], d['counters_after_the_first_calls'], [(s['n_gauss'], s['mean_ms'], s['steady_ms'], s['first8_ms'][:2]) for s in d['segments']])"
    grep "diag step 320[01]" $out/refine_${pool}_${lay//[:.]/_}.err | cut -c1-200
  done
done
for i in 1 2; do timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b$i.json 2> $out/b$i.err; python3 -c "
import json; d=json.loads(open('$out/b$i.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'], d.get('host_step_ms'))"; done
