#!/bin/bash
# round 5, visit AE: the checkpoint grid (64 entries per segment for a list's first 640, 128 behind) -- tests, then layouts
out=gpurun_out/r05_ae
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "heavy_tiles or wide_jobs or clustered or segmented or compact or one_call or learned or cfg2 or full_size" > $out/pytest.log 2>&1
tail -4 $out/pytest.log
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'fill', s.get('fg_bin_emit_sort_capacity'), 'ckpt MB', d['config'].get('seg_ckpt_mb'))" || tail -2 $f.err
}
for lay in uniform clustered:0.5:0.4 clustered:0.8:0.2 needles:0.3:10; do
  for f in 0 640 384 1024; do run fine$f $lay FG_RASTER_SEG_FINE=$f; done
done
