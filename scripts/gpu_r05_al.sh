#!/bin/bash
# round 5, visit AL: the offsets scans inside the columns launch (last workgroup of each group) -- tests, kernel table, bench
out=gpurun_out/r05_al
mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --timeout 900 > $out/pytest.log 2>&1; tail -1 $out/pytest.log
R=$GRAFT_REPO_ROOT
(cd /tmp; rocprofv3 --kernel-trace -d $R/$out/prof -o p -- python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof.json 2> $R/$out/prof.err; python3 $R/scripts/rocprof_top.py $R/$out/prof/p_results.db 11; rm -rf $R/$out/prof)
for i in 1 2 3; do timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b$i.json 2> $out/b$i.err; python3 -c "
import json; d=json.loads(open('$out/b$i.json').read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'])"; done
