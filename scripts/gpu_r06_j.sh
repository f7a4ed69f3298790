#!/bin/bash
# round 6, visit J: an ABSOLUTE cap on a tile's band cost (saturation depth) -- A/B of library builds on the regret layouts
out=gpurun_out/r06_j
mkdir -p $out
export TMPDIR=/tmp
for lib in libfgraster.so libfgraster_cap1024.so libfgraster_cap1536.so libfgraster_cap2560.so; do
  FG_RASTER_LIB=$PWD/freegaussian_amd/$lib REGRET_SETTINGS=auto,even=always timeout 900 python scripts/policy_regret.py $out/regret_$lib.json 2> $out/err_$lib.txt > /dev/null
  python3 - <<PY
import json
b=json.load(open("$out/regret_$lib.json"))
print("$lib", "auto ms:", [round(r["auto_ms"],3) for r in b["rows"]], "sum", round(sum(r["auto_ms"] for r in b["rows"]),3))
PY
done
for lib in libfgraster.so libfgraster_cap1536.so; do
for lay in clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10; do
    FG_RASTER_LIB=$PWD/freegaussian_amd/$lib timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lib $lay', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k})"
done
done
