#!/bin/bash
# round 5, visit AF: backward shares shorter towards the list's front (boundary c(p) = nseg (p / parts)^gamma)
out=gpurun_out/r05_af
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, layout, env...
  local tag=$1 lay=$2; shift 2
  local f=$out/b_${tag}_${lay//[:.+]/_}
  env "$@" timeout 200 python bench.py --layout $lay --steps 32 --warmup 8 --settle-s 0.5 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$tag $lay', 'median', round(d['host_step_ms']['median'],4), 'fwd', s.get('fg_raster_fwd'), 'bwd', s.get('fg_raster_bwd'))" || tail -2 $f.err
}
for G in 4 5 6 8; do
  cd freegaussian_amd/csrc
  touch raster.hip
  make HIPCC="/opt/rocm/bin/hipcc -DFG_SHARE_GAMMA_X4=$G" -j16 > ../../$out/make_$G.log 2>&1
  cd ../..
  for lay in clustered:0.8:0.2 clustered:0.5:0.4 needles:0.3:10 uniform; do run g$G $lay; done
done
