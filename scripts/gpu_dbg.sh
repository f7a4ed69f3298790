#!/bin/bash
out=gpurun_out/${1:-dbg}; mkdir -p $out
FG_PARITY_REPORT=$out/margins.jsonl timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -v "^  File\|pluggy" | tail -40 > $out/pytest.log
tail -25 $out/pytest.log
for sz in "100000 30 480 270" "300000 30 960 540"; do timeout 300 python scripts/model_step_bench.py $sz 2>/dev/null | tr -d '\n ' | cut -c1-900; echo; done
for b in 0 1; do FG_BANDED_BINNING=$b timeout 300 python bench.py --width 960 --height 540 --n-gauss 300000 --steps 50 --warmup 10 --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('960x540 banded=$b', round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms'], d['config']['list_capacity_redos_in_timed_region'])"; done
