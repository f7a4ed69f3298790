python scripts/clustered_check.py 0.5 0.4 > /dev/null 2>&1
for sc in "0.5 0.4" "0.8 0.2"; do
for v in "" ahead16 prefix1024 prefix1024_ahead16; do
for tl in 3072 2048; do
  [ -z "$v" ] && [ $tl = 2048 ] && continue
  [ "$v" = ahead16 ] && [ $tl = 2048 ] && continue
  lib=$PWD/freegaussian_amd/libfgraster${v:+_$v}.so
  echo -n "$sc ${v:-base} tile_len=$tl: "
  FG_HEAVY_TILE_LEN=$tl FG_RASTER_LIB=$lib python scripts/clustered_check.py $sc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print(d['step_ms'], 'fwd', round(d['fwd'],4), 'bwd', round(d['bwd'],4), 'bin', round(s['fg_bin_prepare']+s['fg_bin_emit_sort_capacity'],4))"
done; done; done
