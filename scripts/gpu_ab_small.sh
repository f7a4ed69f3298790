for cfg in "300000 960 540" "100000 480 270"; do set -- $cfg
for env in "FG_RASTER_PPT_BWD=4 FG_RASTER_PPT_FWD=2" "FG_RASTER_PPT_BWD=2 FG_RASTER_PPT_FWD=2" "FG_RASTER_PPT_BWD=1 FG_RASTER_PPT_FWD=1" "FG_RASTER_PPT_BWD=2 FG_RASTER_PPT_FWD=1"; do
env $env python bench.py --n-gauss $1 --width $2 --height $3 --steps 50 --warmup 10 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d[\"stage_ms\"]; print(\"$cfg | $env |\", round(d[\"ms_per_step\"],3), \"ms bwd\", s[\"fg_raster_bwd\"], \"fwd\", s[\"fg_raster_fwd\"])"
done; done
