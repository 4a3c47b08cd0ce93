#!/bin/bash
set -o pipefail
TAG=${1:-r02p}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_decode_float.py tests/test_gpu_parity.py tests/test_gpu_rgb.py -x -q -k "decod or float or k2 or K2 or mirror or cpp" > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -12 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/oneshot_timing.py 2>&1 | tail -9
