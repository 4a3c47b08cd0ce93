#!/bin/bash
# One GPU-box session: parity tests, bench lines, rocprof kernel stats.  Usage: tools_gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $O/${TAG}_bench_cfg2.json 2> $O/${TAG}_bench_cfg2.err || { tail -20 $O/${TAG}_bench_cfg2.err; exit 1; }
cat $O/${TAG}_bench_cfg2.json
timeout -k 10 300 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg4.json 2> $O/${TAG}_bench_cfg4.err || { tail -20 $O/${TAG}_bench_cfg4.err; exit 1; }
cat $O/${TAG}_bench_cfg4.json
timeout -k 10 300 python bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg3.json 2> $O/${TAG}_bench_cfg3.err || { tail -20 $O/${TAG}_bench_cfg3.err; exit 1; }
cat $O/${TAG}_bench_cfg3.json
timeout -k 10 300 python bench.py --workload cfg4iso1 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_cfg4iso1.json 2> $O/${TAG}_bench_cfg4iso1.err || { tail -20 $O/${TAG}_bench_cfg4iso1.err; exit 1; }
cat $O/${TAG}_bench_cfg4iso1.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err || { tail -20 $O/${TAG}_prof.err; exit 1; }
cat $O/${TAG}_prof_bench.json
find $O/${TAG}_prof -name "*stats*" | head
