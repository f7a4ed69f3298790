#!/bin/bash
# round 6, visit K: interleaved 2x2-tile blocks as the XCDs' shares (balance_bands = 3) against cost bands and equal spans
out=gpurun_out/r06_k
mkdir -p $out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "job_lists or heavy or clustered_scene or compact" 2>&1 | tail -3
REGRET_SETTINGS=auto,bands=interleaved,bands=diagonal timeout 900 python scripts/policy_regret.py $out/regret.json $out/regret.md 2> $out/err.txt > /dev/null
python3 - <<PY
import json
b=json.load(open("$out/regret.json"))
for r in b["rows"]: print(r["layout"]["layout"][:40].ljust(42), r["settings_ms"])
PY
for lay in uniform clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10 trained:data/trained_scene_r06.npz; do
  for m in 1 3 4; do
    FG_RASTER_BALANCE=$m timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay balance=$m', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k or 'emit' in k})"
  done
done
