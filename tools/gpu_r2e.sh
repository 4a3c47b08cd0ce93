#!/bin/bash
set -o pipefail
TAG=${1:-r02e}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_q.py -x -q > $O/${TAG}_pytest_q.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest_q.log; tail -8 $O/${TAG}_pytest_q.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/q_stats.py "$@" > $O/${TAG}_stats.txt 2>&1; rc=$?
cat $O/${TAG}_stats.txt
exit $rc
