#!/bin/bash
# round 3 session c: lane-entry slow path (FIC_Q_LANEQ=1) -- parity, then A/B against round 2's slow path (_ab/libfic_hip_old.so)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_q.py tests/test_gpu_rgb_q.py tests/test_gpu_bench_geometry.py -x -q -m gpu > gpurun_out/r03c_pytest_q.log 2>&1
rc=$?; echo "pytest q rc=$rc"; tail -3 gpurun_out/r03c_pytest_q.log
if [ $rc -ne 0 ]; then exit 1; fi
CASES="512,8,8,64 512,8,1,64 512,8,8,1 512,8,1,1 1024,8,8,24 2048,4,1,1 512,8,8,64,S 512,4,8,16 2048,16,8,1 4096,8,1,1 4096,8,8,1"
echo "== new (lane entries)" > gpurun_out/r03c_ab.txt
python tools/q_stats.py $CASES >> gpurun_out/r03c_ab.txt 2>&1
echo "== old (round 2 slow path)" >> gpurun_out/r03c_ab.txt
FIC_HIP_SO=$PWD/_ab/libfic_hip_old.so python tools/q_stats.py $CASES >> gpurun_out/r03c_ab.txt 2>&1
echo "== new again" >> gpurun_out/r03c_ab.txt
python tools/q_stats.py $CASES >> gpurun_out/r03c_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03c_ab.txt | cut -c1-330
