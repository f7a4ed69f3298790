#!/bin/bash
# round 6, visit G: policy regret (before), the GPU gate on the round's code so far
out=gpurun_out/r06_g
mkdir -p $out
export TMPDIR=/tmp
timeout 1500 python scripts/policy_regret.py $out/policy_regret_before.json $out/policy_regret_before.md 16 5 2> $out/regret.err | tail -2
cat $out/policy_regret_before.md | cut -c1-400
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/gate.txt; cat $out/gate.txt
