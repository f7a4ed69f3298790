#!/bin/bash
# round 6, visit D: the reference-schedule run twice from the same seeds (run-to-run spread of the held-out PSNR) + the static run
out=gpurun_out/r06_d
mkdir -p $out
export TMPDIR=/tmp
EV=1000,2000,2900,3000,3500,5000,7000
for r in 1 2; do
  timeout 900 python scripts/train_e2e.py --steps 7000 --eval-at $EV --out $out/e2e_$r > $out/train_$r.log 2>&1
  echo "== run $r"; grep "held-out" $out/train_$r.log
  rm -f $out/e2e_$r/*.ckpt; [ $r = 2 ] && rm -f $out/e2e_$r/trained_scene.npz
done
timeout 900 python scripts/train_e2e.py --steps 7000 --eval-at $EV --warm-up 1000000 --out $out/e2e_static > $out/train_static.log 2>&1
echo "== static"; grep "held-out" $out/train_static.log
rm -f $out/e2e_static/*.ckpt $out/e2e_static/trained_scene.npz
