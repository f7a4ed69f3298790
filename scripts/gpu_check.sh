#!/bin/bash
# One GPU-box visit: parity tests, smoke, headline bench, N-rank control flow, rocprofv3 kernel stats.
# Usage (from the container):  gpurun --timeout 2400 -- 'bash scripts/gpu_check.sh <tag> [skip-tests]'
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
nproc > $out/host.txt; grep -m1 "model name" /proc/cpuinfo >> $out/host.txt; free -g | head -2 >> $out/host.txt
if [ "$2" != "skip-tests" ]; then
  FG_PARITY_REPORT=$out/parity_margins.jsonl timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=12 2>&1 | tail -120 > $out/pytest.log
  timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 > $out/smoke.log
fi
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "rc=$?" >> $out/bench.err
# the launcher: 2 ranks over gloo on the one GPU (control flow), and the RCCL refusal to share a device
FG_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --n-gauss 200000 > $out/dist_gloo.json 2> $out/dist_gloo.err; echo "rc=$?" >> $out/dist_gloo.err
timeout 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $out/dist_nccl.json 2> $out/dist_nccl.err; echo "rc=$?" >> $out/dist_nccl.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $out/prof_bench.json 2> $out/prof.err
cp $out/prof/*/stats_kernel_stats.csv $out/kernel_stats.csv 2>/dev/null || find $out/prof -name "*kernel_stats*" -exec cp {} $out/kernel_stats.csv \;
find $out/prof -name "*kernel_trace*" -delete
cat $out/host.txt; tail -25 $out/pytest.log; cat $out/smoke.log; cut -c1-3000 $out/bench.json; tail -2 $out/bench.err
echo "--- gloo 2 ranks"; tail -2 $out/dist_gloo.err | cut -c1-300; cut -c1-600 $out/dist_gloo.json
echo "--- nccl 2 ranks on 1 GPU (must refuse)"; tail -3 $out/dist_nccl.err | cut -c1-300
head -14 $out/kernel_stats.csv | cut -c1-200
