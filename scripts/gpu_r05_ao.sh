#!/bin/bash
# round 5, visit AO: rocprofv3 kernel tables of the clustered layouts on the final code
out=gpurun_out/r05_ao
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lay in clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10; do
  tag=${lay//[:.+]/_}
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$tag -o stats -- python3 $R/bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag.json 2> $R/$out/prof_$tag.err
  find $R/$out/prof_$tag -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_$tag.csv \;
  rm -rf $R/$out/prof_$tag
  echo "== $lay"; head -14 $R/$out/kernel_stats_$tag.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-60,200-260 | head -14
  python3 -c "
import json; d=json.loads(open('$R/$out/prof_$tag.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['stage_ms'], d.get('host_step_ms',{}).get('median'))"
done
