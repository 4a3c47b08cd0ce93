#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_decode_float.py tests/test_gpu_rgb.py -x -q -m gpu > gpurun_out/r03e_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/r03e_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
python tools/decode_timing.py > gpurun_out/r03e_decode_timing.json 2> gpurun_out/r03e_decode_timing.err
echo rc=$?; cat gpurun_out/r03e_decode_timing.json; tail -3 gpurun_out/r03e_decode_timing.err
