#!/bin/bash
set -o pipefail
TAG=${1:-r02k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_rgb.py tests/test_gpu_decode_float.py tests/test_gpu_parity.py -x -q -k "rgb or decod or K1 or k1 or float" > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -15 $O/${TAG}_pytest.log
exit $rc
