#!/bin/bash
mkdir -p gpurun_out
CASES="512,8,8,64 512,8,1,64 1024,8,8,24 2048,4,1,1 4096,8,1,1 512,8,8,1"
: > gpurun_out/r03d_prio.txt
for v in base prio1 prio4 prio16 base; do
  echo "== $v" >> gpurun_out/r03d_prio.txt
  if [ $v = base ]; then python tools/q_stats.py $CASES >> gpurun_out/r03d_prio.txt 2>&1
  else FIC_HIP_SO=$PWD/_ab/libfic_hip_$v.so python tools/q_stats.py $CASES >> gpurun_out/r03d_prio.txt 2>&1; fi
done
grep -v amdgpu.ids gpurun_out/r03d_prio.txt | cut -c1-140
