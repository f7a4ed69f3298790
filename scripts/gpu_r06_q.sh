#!/bin/bash
# round 6, visit Q: the long-segment switch by the LONGEST segment alone (no count of segments beyond 3072)
out=gpurun_out/r06_q
mkdir -p $out
export TMPDIR=/tmp
REGRET_SETTINGS=auto,long=never,long=always,long_many=off timeout 900 python scripts/policy_regret.py $out/regret.json 2> $out/err.txt > /dev/null
python3 - <<PY
import json
b=json.load(open("$out/regret.json"))
for r in b["rows"]: print(r["layout"]["layout"][:40].ljust(42), r["settings_ms"], r["auto_state"]["long_calls"])
PY
for lay in needles:0.3:10 clustered:0.5:0.4+needles:0.3:10 clustered:0.5:0.4 uniform; do
  for lm in 16 1000000; do
    FG_LONG_MANY=$lm timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay long_many=$lm', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), 'p99', round(d['host_step_ms']['p99'],3), {k:v for k,v in d['stage_ms'].items() if 'emit' in k or 'prepare' in k}, d['config']['long_segment_calls'], d['path_events_in_timed_region']['pool_new_buffers'])"
  done
done
