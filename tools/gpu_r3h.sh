#!/bin/bash
# round 3 session h: k_sweep_q16 (16x16x32 MFMA for 1 isometry at B = 8 / 16) parity + A/B; decoder after the paint / scan rewrite
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests/test_gpu_q.py tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py tests/test_gpu_decode_float.py tests/test_gpu_rgb.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/r03h_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/r03h_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
CASES="512,8,1,64 4096,8,1,1 2048,8,1,4 512,8,1,1 4096,16,1,1 512,8,1,64,S"
for v in shape16 shape32 shape16 shape32; do
  echo "== $v" >> $O/r03h_shape_ab.txt
  if [ $v = shape16 ]; then python tools/q_stats.py $CASES >> $O/r03h_shape_ab.txt 2>&1
  else FIC_HIP_SO=$PWD/_ab/libfic_hip_shape32.so python tools/q_stats.py $CASES >> $O/r03h_shape_ab.txt 2>&1; fi
done
grep -v amdgpu.ids $O/r03h_shape_ab.txt | cut -c1-250
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03h_profd -- python3 tools/decode_timing.py 4096_S > $O/r03h_dec.json 2> $O/r03h_dec.err || { tail -20 $O/r03h_dec.err; exit 1; }
f=$(find $O/r03h_profd -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03h_decode_kernel_stats.csv; rm -rf $O/r03h_profd
cat $O/r03h_dec.json; cut -c1-150 $O/r03h_decode_kernel_stats.csv | head -9
python tools/decode_timing.py > $O/r03h_decode_timing.json 2>/dev/null; cat $O/r03h_decode_timing.json
