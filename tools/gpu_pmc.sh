#!/bin/bash
# PMC passes for the sweep kernel (each counter group in its own rocprofv3 run, kernel-trace only).
# Usage: tools/gpu_pmc.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run_pass() {  # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/${TAG}_pmc_$1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "${@:3}" > $O/${TAG}_pmc_$1.json 2> $O/${TAG}_pmc_$1.err || { tail -5 $O/${TAG}_pmc_$1.err; return 1; }
}
run_pass fetch "FETCH_SIZE" "$@" && run_pass write "WRITE_SIZE" "$@" && \
run_pass sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "$@" && \
run_pass sq2 "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE" "$@" && \
run_pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@"
python3 - <<PY
import csv, glob, collections
for name in ["fetch","write","sq1","sq2","tcc"]:
    fs = glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True)
    for f in fs:
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"]:
                k=(row["Kernel_Name"][:40], row["Counter_Name"]); acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v/n), "launches=%d" % n)
PY
