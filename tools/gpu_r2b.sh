#!/bin/bash
# k_sweep_q bring-up: its own tests first, then the whole parity suite, then the default bench line.
set -o pipefail
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_q.py -x -q > $O/${TAG}_pytest_q.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest_q.log; tail -15 $O/${TAG}_pytest_q.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --cpu-budget 4 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
cat $O/${TAG}_bench_default.json
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_q.py > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -6 $O/${TAG}_pytest.log
exit $rc
