#!/bin/bash
# round 5, visit M: where the wide launch's time goes -- kernel averages per layout / threshold
out=gpurun_out/r05_m
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
prof() {  # tag layout env...
  local tag=$1 lay=$2; shift 2
  for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace -d $R/$out/prof_$tag -o p -- python3 $R/bench.py --layout $lay --steps 32 --warmup 8 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof_$tag.json 2> $R/$out/prof_$tag.err
  for kv in "$@"; do unset "${kv%%=*}"; done
  echo "== $tag $lay $*"
  python3 $R/scripts/rocprof_top.py $R/$out/prof_$tag/p_results.db 4 raster
  rm -rf $R/$out/prof_$tag
}
prof uni_1024 uniform FG_HEAVY_TILE_LEN=1024 FG_HEAVY_TILES=always
prof c54_2560 clustered:0.5:0.4 FG_HEAVY_TILE_LEN=2560
prof c54_1024 clustered:0.5:0.4 FG_HEAVY_TILE_LEN=1024
prof c54_three clustered:0.5:0.4 FG_HEAVY_TILE_LEN=2560 FG_RASTER_HEAVY_WIDE=0
prof c54n_2560 clustered:0.5:0.4+needles:0.3:10 FG_HEAVY_TILE_LEN=2560
