#!/bin/bash
# decoder: float-accumulation tests, RGB + grey decoder parity, timing
mkdir -p gpurun_out
python -m pytest tests/test_gpu_decode_float.py tests/test_gpu_rgb.py tests/test_gpu_parity.py -x -q -m gpu -k "decode or float or rgb or k2 or roundtrip or gui" > gpurun_out/r03m_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/r03m_pytest.log
[ $rc -eq 0 ] || exit 1
python tools/decode_timing.py > gpurun_out/r03m_decode_timing.json 2>/dev/null; cat gpurun_out/r03m_decode_timing.json
