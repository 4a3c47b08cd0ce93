#!/bin/bash
# Round-end GPU session: full parity suite, the default bench line, rocprofv3 kernel stats of that same command,
# FETCH_SIZE / WRITE_SIZE passes for the matrix-core sweep.  Usage: tools/gpu_final.sh <tag>
set -o pipefail
TAG=${1:-r01z}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
cat $O/${TAG}_bench_default.json
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err || { tail -20 $O/${TAG}_prof.err; exit 1; }
find $O/${TAG}_prof -name "*kernel_stats*" | head -2
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  # the default run times the default sweep and, beside it, the matrix-core sweep: both kernels appear in the counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_pmc_$n.json 2> $O/${TAG}_pmc_$n.err || { tail -5 $O/${TAG}_pmc_$n.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for name in ["FETCH_SIZE","WRITE_SIZE","SQ_INSTS_VALU"]:
    for f in glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"]:
                k=(row["Kernel_Name"][:40], row["Counter_Name"]); acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v/n), "launches=%d" % n)
PY
