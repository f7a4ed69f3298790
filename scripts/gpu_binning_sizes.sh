#!/bin/bash
# The three binning paths at heavier sizes (mean supertile segment 3000+ entries: the LDS-capacity classes of
# csrc/stbin.hip's sort).  Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_binning_sizes.sh <tag>'
tag=${1:-binsizes}
out=gpurun_out/$tag
mkdir -p $out
for cfg in "3000000 1920 1080" "2000000 2560 1440" "4000000 3840 2160"; do
  set -- $cfg
  for b in depthfirst supertile; do
    FG_BINNING=$b timeout 400 python bench.py --steps 20 --warmup 5 --settle-s 0.3 --no-cpu-baseline --no-graph --n-gauss $1 --width $2 --height $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('$b'.ljust(11), 'N=$1 $2x$3', 'step %.4f prepare %.4f emit_sort %.4f fwd %.4f bwd %.4f I=%d I_raster=%d' % (d['ms_per_step'], s['fg_bin_prepare'], s['fg_bin_emit_sort_capacity'], s['fg_raster_fwd'], s['fg_raster_bwd'], d['config']['I'], d['config']['I_raster']))" | tee -a $out/sizes.txt
  done
done
