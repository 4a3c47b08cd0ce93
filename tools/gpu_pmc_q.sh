#!/bin/bash
# SQ / GRBM counters of the sweep kernel for a q_stats case (default: the bench workload).  Usage: tools/gpu_pmc_q.sh <tag> <case>
set -o pipefail
TAG=${1:-q}; CASE=${2:-512,8,8,64}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 300 python tools/q_stats.py $CASE 2>&1 | tail -3
export TMPDIR=/tmp; cd /tmp
run_pass() {
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/${TAG}_pmc_$1 -- python3 $R/tools/q_stats.py $CASE > $O/${TAG}_pmc_$1.txt 2> $O/${TAG}_pmc_$1.err || { tail -5 $O/${TAG}_pmc_$1.err; return 1; }
}
run_pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" && \
run_pass b "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" && \
run_pass c "FETCH_SIZE" && run_pass d "WRITE_SIZE" && run_pass e "TCC_HIT_sum TCC_MISS_sum"
python3 - <<PY
import csv, glob, collections
for name in ["a","b","c","d","e"]:
    for f in glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"]:
                k=(row["Kernel_Name"][:34], row["Counter_Name"]); acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v/n), "launches=%d" % n)
PY
