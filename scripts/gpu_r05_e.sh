#!/bin/bash
# round 5, visit E: sparse payloads of the view-DP exchange (lockstep tests, bytes per layout), bench launcher over gloo
out=gpurun_out/r05_e
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 900 -x -k "view_dp or world_size_8 or factored" 2>&1 | tail -15 > $out/pytest_dp.log
tail -40 $out/pytest_dp.log | cut -c1-600
timeout 600 python scripts/dp_payload_bytes.py > $out/dp_payload_bytes.json 2> $out/dp_payload_bytes.err; tail -c 2500 $out/dp_payload_bytes.json; tail -2 $out/dp_payload_bytes.err
FG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --n-gauss 200000 > $out/dist_gloo.json 2> $out/dist_gloo.err; python3 -c "
import json; d=json.load(open('$out/dist_gloo.json')); print(d.get('per_rank_device'), d.get('distinct_devices'), d.get('per_rank_mpix_per_s'))"
