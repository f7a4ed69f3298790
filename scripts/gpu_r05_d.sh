#!/bin/bash
# round 5, visit D: pose gradient + footprint tests, whole suite, kernel-level times of the binning with masks on / off
out=gpurun_out/r05_d
mkdir -p $out
export TMPDIR=/tmp
FG_PARITY_REPORT=$out/parity_margins.jsonl timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40 > $out/pytest.log
grep -E "passed|failed|FAILED|Error" $out/pytest.log | head -20
for exact in 1 0; do
for lay in uniform needles:0.3:10; do
  tag=${lay//[:.+]/_}_exact$exact
  FG_EXACT_TILES=$exact timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -o stats -- python3 bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/prof_$tag.json 2> $out/prof_$tag.err
  find $out/prof_$tag -name "*kernel_stats*" -exec cp {} $out/kernel_stats_$tag.csv \;
  rm -rf $out/prof_$tag
  echo "== $tag"; head -14 $out/kernel_stats_$tag.csv | cut -d, -f1-4 | cut -c1-150
done; done
