python -m pytest tests -m gpu -x -q -k "compact_checkpoint or segmented_backward or heavy_tiles or job_lists or one_call or clustered" 2>&1 | grep -v "^  File \"/usr" | tail -5
for i in 1 2; do
for c in 1 0; do
FG_COMPACT_SLOTS=$c python bench.py --no-clustered --steps 100 2>/dev/null | tail -1 > gpurun_out/bench_ab_${c}_$i.json
python - <<PY
import json
d=json.load(open('gpurun_out/bench_ab_${c}_$i.json'))
print('compact=$c', round(d['value'],1), round(d['ms_per_step'],4), round(d['hip_event_mpix_per_s'],1), d['stage_ms'], d['host_step_ms']['median'])
PY
done
done
