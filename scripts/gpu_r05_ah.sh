#!/bin/bash
# round 5, visit AH: the long segments' sample / count / scatter launches beside the small sort -- list tests, then layouts
out=gpurun_out/r05_ah
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -q --timeout 600 -k "long_segments or clustered or stbin or supertile or graph or binning" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'), 'prepare', s.get('fg_bin_prepare'), 'fill', s.get('fg_bin_emit_sort_capacity'))" || tail -2 $f.err
}
for lay in clustered:0.5:0.4 clustered:0.8:0.2 clustered:0.5:0.4+needles:0.3:10; do
  run serial $lay FG_LONG_OVERLAP=0
  run beside $lay
  run serial $lay FG_LONG_OVERLAP=0
  run beside $lay
done
