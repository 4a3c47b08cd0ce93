#!/bin/bash
# full GPU suite, then part D of the evidence (bench line, single-image kernel stats, traffic stamp)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
TAG=${1:-r03I}
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_final_r3.sh D $TAG
