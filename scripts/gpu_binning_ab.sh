#!/bin/bash
# The three binning paths: parity tests, step time A/B, per-kernel durations of the default.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_binning_ab.sh <tag> [full]'
tag=${1:-bin}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
sel='-k "supertile or binning or footprint or cfg1 or full_pipeline or speculative or redoes or 65536 or graph"'
[ "$2" = "full" ] && sel=""
eval FG_PARITY_REPORT=$out/margins.jsonl timeout 1200 python -m pytest tests -m gpu -q --timeout 600 $sel 2>&1 | grep -v "^  File\|pluggy" | tail -40 > $out/pytest.log
tail -12 $out/pytest.log
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1  # (the first process on a fresh box runs slow)
for b in depthfirst supertile supertile; do
  FG_BINNING=$b timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $out/bench_$b.json 2> $out/bench_$b.err
  python - <<PY
import json
try:
    d=json.load(open("$out/bench_$b.json"))
    print("$b", round(d["value"],1), "Mpix/s", round(d["ms_per_step"],4), "ms", d["stage_ms"])
except Exception as e:
    print("$b failed", e); print(open("$out/bench_$b.err").read()[-1500:])
PY
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o stats -- python3 bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > $out/prof_bench.json 2> $out/prof.err
find $out/prof -name "*kernel_stats*" -exec cp {} $out/kernel_stats.csv \;
find $out/prof -name "*kernel_trace*" -delete
python - <<PY
import csv
for r in list(csv.DictReader(open("$out/kernel_stats.csv")))[:13]:
    print(f'{r["Name"][:64]:64s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY
