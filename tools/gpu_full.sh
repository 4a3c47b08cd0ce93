#!/bin/bash
# Full GPU validation: whole parity suite, default bench line, self-launched 2-rank bench (gloo rehearsal on one GPU).
set -o pipefail
TAG=${1:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -8 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
cat $O/${TAG}_bench_default.json
FIC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 > $O/${TAG}_bench_gloo2.json 2> $O/${TAG}_bench_gloo2.err || { tail -20 $O/${TAG}_bench_gloo2.err; exit 1; }
cat $O/${TAG}_bench_gloo2.json
