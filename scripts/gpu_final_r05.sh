#!/bin/bash
# End-of-round visit, round 5: the driver's bench command on the cold box first, then scripts/gpu_check.sh (suite, smoke,
# bench, launcher, rocprofv3 kernel stats), the PMC passes, the step series across refinements, the long fuzz sweeps.
tag=${1:-r05_final}
out=gpurun_out/$tag
mkdir -p $out
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/driver_command_cold.json 2> $out/driver_command_cold.err
python3 -c "
import json; d=json.loads(open('$out/driver_command_cold.json').read().strip().splitlines()[-1]); print('driver command, cold box:', round(d['value'],1), 'Mpix/s', round(d['ms_per_step'],4), 'ms; cpu_baseline grad max', d['cpu_baseline']['grad_rel_l2_hip_vs_oracle_max'], 'full frame', d['cpu_full_frame']['grad_rel_l2_hip_vs_oracle_max'], 'graphed', d.get('graphed',{}).get('ms_per_step')); print({k:(v.get('ms_per_step_median'), v.get('error')) for k,v in d.get('clustered_layouts',{}).items()})"
bash scripts/gpu_check.sh $tag
bash scripts/gpu_pmc.sh ${tag}_pmc > gpurun_out/${tag}_pmc.txt 2>&1
tail -8 gpurun_out/${tag}_pmc.txt | cut -c1-160
for lay in uniform clustered:0.5:0.4; do
  timeout 600 python scripts/refine_step_bench.py $lay > $out/refine_${lay//[:.]/_}.json 2> $out/refine_${lay//[:.]/_}.err
  python3 -c "
import json; d=json.load(open('$out/refine_${lay//[:.]/_}.json')); print('$lay', 'series / steady', d['series_mean_over_steady_state'], d['counters_after_the_first_calls'], [(s['n_gauss'], s['mean_ms'], s['steady_ms']) for s in d['segments']])"
done
timeout 1500 python scripts/fuzz_parity.py 120 11 > $out/fuzz_small.txt 2>&1; tail -1 $out/fuzz_small.txt
timeout 1500 python scripts/fuzz_parity.py 24 4 big > $out/fuzz_big.txt 2>&1; tail -1 $out/fuzz_big.txt
