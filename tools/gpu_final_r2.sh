#!/bin/bash
# Round-2 evidence session: full parity suite, default bench line, rocprofv3 kernel stats + SQ counters of that command,
# FETCH/WRITE traffic (-> traffic.json), the sweep matrix of every configuration, one-shot / RGB / in-process timings.
set -o pipefail
TAG=${1:-r02z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
echo "bench default ok"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
[ -n "$SKIP_MATRIX" ] || timeout -k 10 600 python tools/q_stats.py 512,8,8,64 512,8,8,64,U,3 512,8,8,64,U,5 512,8,8,64,S 512,8,1,64 512,8,1,64,U,3 512,8,8,1 512,8,8,1,U,3 \
  2048,4,1,1 2048,4,1,1,U,3 2048,4,1,1,U,2 4096,8,8,1 4096,8,8,1,U,3 4096,8,1,1 4096,8,1,1,U,3 1024,8,8,24 1024,8,8,24,U,3 512,4,8,64 512,4,8,64,U,3 \
  4096,16,1,1 4096,16,1,1,U,3 4096,16,8,1 4096,16,8,1,U,3 > $O/${TAG}_sweep_matrix.txt 2>&1 || { tail $O/${TAG}_sweep_matrix.txt; exit 1; }
[ -n "$SKIP_MATRIX" ] || cat $O/${TAG}_sweep_matrix.txt
timeout -k 10 300 python tools/oneshot_timing.py > $O/${TAG}_oneshot.json 2> $O/${TAG}_oneshot.err || { tail $O/${TAG}_oneshot.err; exit 1; }
timeout -k 10 300 python tools/rgb_timing.py > $O/${TAG}_rgb.json 2> $O/${TAG}_rgb.err || { tail $O/${TAG}_rgb.err; exit 1; }
FIC_FAKE_DEVICES=8 timeout -k 10 300 python bench.py --inproc --gpus 4 --workload cfg4 --size 2048 --steps 3 --warmup 1 > $O/${TAG}_inproc4_fake.json 2> $O/${TAG}_inproc.err || { tail $O/${TAG}_inproc.err; exit 1; }
timeout -k 10 300 python bench.py --inproc --gpus 1 --workload cfg4 --size 2048 --steps 3 --warmup 1 > $O/${TAG}_inproc1.json 2>> $O/${TAG}_inproc.err || { tail $O/${TAG}_inproc.err; exit 1; }
FIC_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 > $O/${TAG}_bench_gloo2.json 2> $O/${TAG}_bench_gloo2.err || { tail -20 $O/${TAG}_bench_gloo2.err; exit 1; }
echo "timings ok"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt > $O/${TAG}_prof_bench.json 2> $O/${TAG}_prof.err || { tail -20 $O/${TAG}_prof.err; exit 1; }
run_pass() {
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/${TAG}_pmc_$1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/${TAG}_pmc_$1.json 2> $O/${TAG}_pmc_$1.err || { tail -5 $O/${TAG}_pmc_$1.err; return 1; }
}
run_pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" && \
run_pass b "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" && \
run_pass e "TCC_HIT_sum TCC_MISS_sum" || exit 1
python3 - <<PY > $O/${TAG}_pmc_summary.txt
import csv, glob, collections
print("rocprofv3 --pmc passes over: bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt (cfg2: 64 x 512x512, B=8, 8 iso); per-launch averages of the sweep kernel")
for name in ["a","b","e"]:
    for f in glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"]:
                k=(row["Kernel_Name"][:34], row["Counter_Name"]); acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v/n), "launches=%d" % n)
PY
cat $O/${TAG}_pmc_summary.txt
cd $R && bash tools/gpu_traffic.sh ${TAG}
