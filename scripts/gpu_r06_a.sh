#!/bin/bash
# round 6, visit A: the end-to-end tests (twin + short run), then the full 7000-step run on the reference's schedule
out=gpurun_out/r06_a
mkdir -p $out
export TMPDIR=/tmp
FG_TWIN_REPORT=$out/twin.json timeout 900 python -m pytest tests/test_e2e.py -m gpu -x -q -s 2>&1 | tail -15 > $out/test_e2e.txt
cat $out/test_e2e.txt
timeout 1500 python scripts/train_e2e.py --steps 7000 --out $out/e2e > $out/train.log 2>&1
tail -30 $out/train.log
rm -f $out/e2e/*.ckpt   # (hundreds of MB: stays on the box)
ls -la $out/e2e
