#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, rocprofv3 kernel stats.
# Usage (from the container):  gpurun --timeout 2400 -- 'bash scripts/gpu_check.sh <tag>'
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
nproc > $out/host.txt; grep -m1 "model name" /proc/cpuinfo >> $out/host.txt; free -g | head -2 >> $out/host.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 300 --durations=8 2>&1 | tail -40 > $out/pytest.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 > $out/smoke.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc=$?" >> $out/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $out/prof_bench.json 2> $out/prof.err
cp $out/prof/*/stats_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/prof -name "*kernel_stats*" -exec cp {} $out/kernel_stats.csv \;
find $out/prof -name "*kernel_trace*" -delete
cat $out/host.txt; tail -8 $out/pytest.log; cat $out/smoke.log; cat $out/bench.json; tail -2 $out/bench.err
head -12 $out/kernel_stats.csv | cut -c1-200
