#!/bin/bash
out=gpurun_out/r03_lowres_mm; mkdir -p $out
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-graph > /dev/null 2>&1
for cfg in "100000 480 270" "50000 320 192" "150000 640 360" "200000 480 270"; do
  set -- $cfg
  for v in base mm100 "mm100 FG_RASTER_SEG_PARTS=10" "mm100 FG_RASTER_SEG_PARTS=16"; do
    set -- $cfg
    lib=""; e=""
    case "$v" in mm100*) lib="FG_RASTER_LIB=$PWD/freegaussian_amd/libfg_mm100.so";; esac
    case "$v" in *PARTS*) e="${v#mm100 }";; esac
    env $lib $e timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --n-gauss $1 --width $2 --height $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']
print('N=$1 $2x$3 [$v]', 'eager %.4f graphed %.4f' % (d['ms_per_step'], d.get('graphed',{}).get('ms_per_step', float('nan'))), 'sum %.4f' % sum(s.values()), {k.replace('fg_',''): v for k, v in s.items() if 'raster' in k})" | tee -a $out/mm.txt
  done
done
