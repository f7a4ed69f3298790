#!/bin/bash
# round 6, visit R: footprint masks where they pay + long segments by the longest segment alone: the gate, the bench line
out=gpurun_out/r06_r
mkdir -p $out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $out/gate.txt; cat $out/gate.txt | cut -c1-300
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.loads([l for l in open('$out/bench_default.json').read().strip().splitlines() if l.startswith('{')][-1])
print('default:', round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms'], d['path_events_in_timed_region'], d['config']['footprint_masks'])
for k,v in d['clustered_layouts'].items(): print(k, v.get('ms_per_step'), v.get('ms_per_step_median'), v.get('host_step_ms_p99'), v.get('stage_ms'), v.get('error'))"
