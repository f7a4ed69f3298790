#!/bin/bash
# round 6, visit P: the uneven-scene policy (interleaved shares + finer thresholds): regret table after, gate, default bench line
out=gpurun_out/r06_p
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python scripts/policy_regret.py $out/policy_regret_after.json $out/policy_regret_after.md 16 5 2> $out/regret.err | tail -1
python3 - <<PY
import json
b=json.load(open("$out/policy_regret_after.json"))
for r in b["rows"]: print(r["layout"]["layout"][:40].ljust(42), "auto", round(r["auto_ms"],3), "best", r["best"].ljust(16), round(r["best_ms"],3), "regret", round(r["regret"],3), "r05:", r["settings_ms"].get("uneven=r05"))
PY
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $out/gate.txt; cat $out/gate.txt
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.loads([l for l in open('$out/bench_default.json').read().strip().splitlines() if l.startswith('{')][-1])
print('default:', round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms'], d['path_events_in_timed_region'])
for k,v in d['clustered_layouts'].items(): print(k, v.get('ms_per_step'), v.get('ms_per_step_median'), v.get('host_step_ms_p99'), v.get('stage_ms'), v.get('path_events_in_timed_region'), v.get('error'))"
