#!/bin/bash
# End-of-round visit, round 6: the driver's bench command on the cold box first, then scripts/gpu_check.sh (suite, smoke,
# bench, launcher, rocprofv3 kernel stats), the PMC passes (bench scene and trained scene), kernel tables of the layouts, the
# step series across refinements, the end-to-end run on HEAD.
tag=${1:-r06_final}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
T=data/trained_scene_r06.npz
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_command_cold.json 2> $out/driver_command_cold.err
python3 -c "
import json; d=json.loads([l for l in open('$out/driver_command_cold.json').read().strip().splitlines() if l.startswith('{')][-1]); print('driver command, cold box:', round(d['value'],1), 'Mpix/s', round(d['ms_per_step'],4), 'ms; cpu_baseline grad max', d['cpu_baseline']['grad_rel_l2_hip_vs_oracle_max'], 'full frame', d['cpu_full_frame']['grad_rel_l2_hip_vs_oracle_max'], 'graphed', d.get('graphed',{}).get('ms_per_step'), d['host_step_ms'], d['path_events_in_timed_region']); print({k:(v.get('ms_per_step'), v.get('ms_per_step_median'), v.get('host_step_ms_p99'), v.get('error')) for k,v in d.get('clustered_layouts',{}).items()})"
bash scripts/gpu_check.sh $tag
bash scripts/gpu_pmc.sh ${tag}_pmc > gpurun_out/${tag}_pmc.txt 2>&1
tail -8 gpurun_out/${tag}_pmc.txt | cut -c1-160
cd /tmp
for lay in clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10 trained:$R/$T; do
  tag2=$(echo $lay | sed 's/[:.+\/]/_/g' | cut -c1-24)
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_$tag2 -o stats -- python3 $R/bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag2.json 2> $R/$out/prof_$tag2.err
  find $R/$out/prof_$tag2 -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_$tag2.csv \;
  rm -rf $R/$out/prof_$tag2
  echo "== $lay"; python3 - <<PY
import csv
rows=list(csv.reader(open("$R/$out/kernel_stats_$tag2.csv")))
for r in rows[1:14]: print("  ", r[0].replace("(anonymous namespace)::","")[:44].ljust(46), r[1].rjust(6), round(float(r[3])/1e3,1))
PY
done
cd $R
for lay in uniform clustered:0.5:0.4; do
  timeout 600 python scripts/refine_step_bench.py $lay > $out/refine_${lay//[:.]/_}.json 2> $out/refine_${lay//[:.]/_}.err
  python3 -c "
import json; d=json.load(open('$out/refine_${lay//[:.]/_}.json')); print('$lay', 'series / steady', d['series_mean_over_steady_state'], d['counters_after_the_first_calls'], [(s['n_gauss'], s['mean_ms'], s['steady_ms']) for s in d['segments']])"
done
timeout 900 python scripts/train_e2e.py --steps 7000 --eval-at 1000,2000,2900,3000,3500,5000,7000 --out $out/e2e > $out/train.log 2>&1
grep "held-out" $out/train.log
rm -f $out/e2e/*.ckpt $out/e2e/trained_scene.npz
timeout 900 python scripts/fuzz_parity.py 120 11 > $out/fuzz_small.txt 2>&1; tail -1 $out/fuzz_small.txt
