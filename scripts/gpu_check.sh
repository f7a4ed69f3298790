#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, rocprofv3 kernel stats, PPT sweep.
# Usage (from the container):  gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh <tag>'
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 180 2>&1 | tail -40 > $out/pytest.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 > $out/smoke.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc=$?" >> $out/bench.err
for ppt in 1 2 4; do
  FG_RASTER_PPT=$ppt timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_ppt$ppt.json 2>/dev/null
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err
find $out/prof -name "*kernel_stats*" -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/prof/*/*kernel_trace* 2>/dev/null
tail -15 $out/pytest.log; cat $out/smoke.log; cat $out/bench.json; tail -2 $out/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_ppt*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), d["ms_per_step"], d["stage_ms"])
    except Exception as e: print(f, "ERR", e)
PY
head -30 $out/kernel_stats.csv
