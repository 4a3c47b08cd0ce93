#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/r03b_2rank.err
for i in 1 2 3 4 5 6 7 8; do
FIC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 2 --warmup 1 --extra-size 512 --extra-steps 2 > gpurun_out/r03b_2rank_$i.json 2>> gpurun_out/r03b_2rank.err || { echo "run $i rc=$?"; }
done
grep "verify" gpurun_out/r03b_2rank.err | head -20
echo done
