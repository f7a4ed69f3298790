true
FG_AB_SKIP_TESTS=1 bash scripts/gpu_ab.sh r04_balance "FG_RASTER_BALANCE=1" "FG_RASTER_BALANCE=0" "FG_RASTER_BALANCE=100" "FG_RASTER_BALANCE=1" "FG_RASTER_BALANCE=0"
for sc in "0.5 0.4" "0.8 0.2"; do for st in 1 0; do FG_RASTER_BALANCE=$st python scripts/clustered_check.py $sc 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('balance=$st', d['scene'], d['step_ms'], 'fwd', round(d['fwd'],3), 'bwd', round(d['bwd'],3))"; done; done
