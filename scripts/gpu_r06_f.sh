#!/bin/bash
# round 6, visit F: sb_count with dealt (Gaussian, supertile row) items: binning tests, kernel tables on three scenes
out=gpurun_out/r06_f
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=data/trained_scene_r06.npz
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning or footprint or supertile or long_seg or job_lists or clustered_1m or randomised" 2>&1 | tail -5 > $out/tests.txt; cat $out/tests.txt
cd /tmp
for lay in uniform needles:0.3:10 trained:$R/$T; do
  tag=$(echo $lay | sed 's/[:.+\/]/_/g' | cut -c1-24)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$tag -o stats -- python3 $R/bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag.json 2> $R/$out/prof_$tag.err
  find $R/$out/prof_$tag -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_$tag.csv \;
  rm -rf $R/$out/prof_$tag
  echo "== $lay"
  python3 - <<PY
import csv, json
rows=list(csv.reader(open("$R/$out/kernel_stats_$tag.csv")))
for r in rows[1:15]: print("  ", r[0].replace("(anonymous namespace)::","")[:44].ljust(46), r[1].rjust(6), round(float(r[3])/1e3,1))
d=json.loads([l for l in open("$R/$out/prof_$tag.json").read().strip().splitlines() if l.startswith("{")][-1]); print("  step", round(d["ms_per_step"],4))
PY
done
