#!/bin/bash
# A/B of environment knobs.  Usage: gpurun -- 'bash scripts/gpu_ab.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...'
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
[ -n "$FG_AB_SKIP_TESTS" ] || timeout 900 python -m pytest tests -m gpu -q --timeout 300 -x 2>&1 | tail -5 > $out/pytest.log
tail -3 $out/pytest.log
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 200 python bench.py --steps 60 --warmup 16 --no-cpu-baseline --no-graph > $out/ab_$i.json 2>/dev/null
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/ab_$i.json").read().strip().splitlines()[-1]); s=d["stage_ms"]
    print("$cfg", "|", round(d["value"],1), "Mpix/s", round(d["ms_per_step"],3), "ms", {k:v for k,v in s.items() if v>0.03})
except Exception as e: print("$cfg", "ERR", e)
PY
done
