#!/bin/bash
# round 6, visit U: the reference's FULL schedule -- 30 000 steps (densify to 15 000, cull-only after, opacity resets every 3000)
out=gpurun_out/r06_u
mkdir -p $out
export TMPDIR=/tmp
timeout 2400 python scripts/train_e2e.py --steps 30000 --eval-at 1000,3000,7000,15000,20000,30000 --out $out/e2e > $out/train.log 2>&1
grep "held-out" $out/train.log; tail -3 $out/train.log | cut -c1-1500
rm -f $out/e2e/*.ckpt $out/e2e/trained_scene.npz
