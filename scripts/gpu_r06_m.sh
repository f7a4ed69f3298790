#!/bin/bash
# round 6, visit M: long tiles first in the job lists (A/B of builds) x the XCDs' shares (cost bands / interleaved blocks)
out=gpurun_out/r06_m
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "job_lists or heavy or clustered or compact or segment or learned or wide" 2>&1 | tail -3
for lay in uniform clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10 trained:data/trained_scene_r06.npz; do
  for lib in libfgraster_lf0.so libfgraster.so; do
  for m in 1 3; do
    FG_RASTER_LIB=$PWD/freegaussian_amd/$lib FG_RASTER_BALANCE=$m timeout 300 python bench.py --layout $lay --steps 48 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/b.json 2> $out/b.err
    python3 -c "
import json; d=json.loads([l for l in open('$out/b.json').read().strip().splitlines() if l.startswith('{')][-1]); print('$lay $lib balance=$m', round(d['ms_per_step'],4), 'median', round(d['host_step_ms']['median'],4), {k:v for k,v in d['stage_ms'].items() if 'raster' in k})"
  done
  done
done
