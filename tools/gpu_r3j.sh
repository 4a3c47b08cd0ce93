#!/bin/bash
# round 3 session j: folded sweep with the cheap upper-bound test first (A/B), decoder with the maps walked out of LDS
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests/test_gpu_q.py tests/test_gpu_bench_geometry.py tests/test_gpu_d4.py tests/test_gpu_decode_float.py -x -q -m gpu > $O/r03j_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/r03j_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
CASES="512,8,8,64 1024,8,8,24 512,8,8,64,S 2048,16,8,1 512,8,8,1 4096,8,8,1"
rm -f $O/r03j_ubound_ab.txt
for v in ubound noubound ubound noubound; do
  echo "== $v" >> $O/r03j_ubound_ab.txt
  if [ $v = ubound ]; then python tools/q_stats.py $CASES >> $O/r03j_ubound_ab.txt 2>&1
  else FIC_HIP_SO=$PWD/_ab/libfic_hip_noubound.so python tools/q_stats.py $CASES >> $O/r03j_ubound_ab.txt 2>&1; fi
done
grep -v amdgpu.ids $O/r03j_ubound_ab.txt | cut -c1-250
python tools/decode_timing.py > $O/r03j_decode_timing.json 2>/dev/null; cat $O/r03j_decode_timing.json
