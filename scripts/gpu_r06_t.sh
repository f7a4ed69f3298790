#!/bin/bash
# round 6, visit T: the long-segment chain in two launches (count + scatter in one pass into slabs; overflow in the bucket
# sort's last workgroup): binning tests incl. the forced-overflow pass, kernel tables of three uneven scenes
out=gpurun_out/r06_t
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning or long_seg or clustered_1m or supertile or randomised or heavy or clustered_scene" 2>&1 | tail -5
cd /tmp
for lay in clustered:0.5:0.4 clustered:0.8:0.2 trained:$R/data/trained_scene_r06.npz; do
  tag2=$(echo $lay | sed 's/[:.+\/]/_/g' | cut -c1-24)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$tag2 -o stats -- python3 $R/bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag2.json 2> $R/$out/prof_$tag2.err
  find $R/$out/prof_$tag2 -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_$tag2.csv \;
  rm -rf $R/$out/prof_$tag2
  echo "== $lay"; python3 - <<PY
import csv, json
rows=list(csv.reader(open("$R/$out/kernel_stats_$tag2.csv")))
for r in rows[1:14]: print("  ", r[0].replace("(anonymous namespace)::","")[:44].ljust(46), r[1].rjust(6), round(float(r[3])/1e3,1))
d=json.loads([l for l in open("$R/$out/prof_$tag2.json").read().strip().splitlines() if l.startswith("{")][-1]); print("  step", round(d["ms_per_step"],4), "median", round(d["host_step_ms"]["median"],4))
PY
done
