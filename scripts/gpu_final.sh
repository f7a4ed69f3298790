#!/bin/bash
# End-of-round visit: the driver's bench command on the cold box first, then scripts/gpu_check.sh, then the PMC passes.
tag=${1:-r04_final}
mkdir -p gpurun_out/$tag
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/driver_command_cold.json 2> gpurun_out/$tag/driver_command_cold.err
python3 -c "
import json; d=json.load(open('gpurun_out/$tag/driver_command_cold.json')); print('driver command, cold box:', round(d['value'],1), 'Mpix/s', round(d['ms_per_step'],4), 'ms; cpu_baseline grad max', d['cpu_baseline']['grad_rel_l2_hip_vs_oracle_max'], 'full frame', d['cpu_full_frame']['grad_rel_l2_hip_vs_oracle_max'], 'graphed', d.get('graphed',{}).get('ms_per_step'))"
bash scripts/gpu_check.sh $tag
bash scripts/gpu_pmc.sh ${tag}_pmc > gpurun_out/${tag}_pmc.txt 2>&1
tail -25 gpurun_out/${tag}_pmc.txt | cut -c1-200
