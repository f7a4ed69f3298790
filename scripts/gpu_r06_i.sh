#!/bin/bash
# round 6, visit I: measured band costs: tests, regret table after, clustered bench lines
out=gpurun_out/r06_i
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "measured_band or heavy or clustered or learned or compact or segment or randomised or job_lists" 2>&1 | tail -8 > $out/tests.txt; cat $out/tests.txt
timeout 1500 python scripts/policy_regret.py $out/policy_regret_walks.json $out/policy_regret_walks.md 16 5 2> $out/regret.err | tail -1
python3 - <<PY
import json
a=json.load(open("gpurun_out/r06_g/policy_regret_before.json")) if __import__("os").path.exists("gpurun_out/r06_g/policy_regret_before.json") else None
b=json.load(open("$out/policy_regret_walks.json"))
for i,r in enumerate(b["rows"]):
    print(r["layout"]["layout"][:40].ljust(42), "auto", round(r["auto_ms"],3), "best", r["best"].ljust(12), round(r["best_ms"],3), "regret", round(r["regret"],3), "even=always", r["settings_ms"].get("even=always"))
PY
for lay in clustered:0.5:0.4 clustered:0.8:0.2 trained:data/trained_scene_r06.npz uniform; do
  for w in 1 0; do
    FG_TILE_WALKS=$w timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay walks=$w', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k})"
  done
done
