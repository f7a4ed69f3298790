#!/bin/bash
# Exercise bench.py's N>1 control flow on the 1-GPU box: 2 ranks sharing the GPU over gloo.
out=gpurun_out/${1:-dist}
mkdir -p $out
FG_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --n-gauss 200000 > $out/dist.json 2> $out/dist.err
echo "rc=$?"; tail -3 $out/dist.err | cut -c1-300; cat $out/dist.json | cut -c1-900
