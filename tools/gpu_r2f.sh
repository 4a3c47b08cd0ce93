#!/bin/bash
# stats for selected cases + rocprofv3 kernel stats of the default bench command
set -o pipefail
TAG=${1:-r02f}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 600 python tools/q_stats.py "$@" > $O/${TAG}_stats.txt 2>&1 || { cat $O/${TAG}_stats.txt; exit 1; }
cat $O/${TAG}_stats.txt
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err || { tail -20 $O/${TAG}_prof.err; exit 1; }
f=$(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1); cat $f | cut -d, -f1-8 | head -20
