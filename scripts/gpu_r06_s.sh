#!/bin/bash
# round 6, visit S: the trained frame against the oracle (in the short end-to-end test); the regret table on HEAD
out=gpurun_out/r06_s
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_e2e.py -m gpu -x -q -s 2>&1 | tail -12 | cut -c1-400
timeout 1500 python scripts/policy_regret.py $out/policy_regret_head.json $out/policy_regret_head.md 16 5 2> $out/regret.err | tail -1
python3 - <<PY
import json
b=json.load(open("$out/policy_regret_head.json"))
for r in b["rows"]: print(r["layout"]["layout"][:40].ljust(42), "auto", round(r["auto_ms"],3), "best", r["best"].ljust(16), round(r["best_ms"],3), "regret", round(r["regret"],3))
PY
