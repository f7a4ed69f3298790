#!/bin/bash
# PPT sweep for raster fwd/bwd.  Usage: gpurun -- 'bash scripts/gpu_sweep.sh <tag>'
tag=${1:-sweep}
out=gpurun_out/$tag
mkdir -p $out
for f in 1 2 4; do for b in 1 2 4; do
  FG_RASTER_PPT_FWD=$f FG_RASTER_PPT_BWD=$b timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/b_${f}_${b}.json 2>/dev/null
done; done
python3 - <<PY
import json,glob
for fn in sorted(glob.glob("$out/b_*.json")):
    try:
        d=json.loads(open(fn).read().strip().splitlines()[-1]); s=d["stage_ms"]
        print(fn.split("/")[-1], round(d["value"],1), round(d["ms_per_step"],3), "fwd", s["fg_raster_fwd"], "bwd", s["fg_raster_bwd"])
    except Exception as e: print(fn, "ERR", e)
PY
