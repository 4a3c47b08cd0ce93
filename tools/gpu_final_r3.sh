#!/bin/bash
# Round-3 evidence sessions (gpurun allows 20 minutes per call, hence three parts):
#   tools/gpu_final_r3.sh A <tag>   full parity suite, default bench line, smoke, sweep matrix, one-shot / RGB / decoder timings, rehearsals
#   tools/gpu_final_r3.sh C <tag>   RGB timings, the 4-rank rehearsal of the default bench (gloo, one GPU), fuzz soak of every sweep kernel
#   tools/gpu_final_r3.sh D <tag>   after a late change: bench line + single-image kernel stats + traffic stamp only
#   tools/gpu_final_r3.sh B <tag>   rocprofv3 kernel stats + SQ / TCC counters of the default bench command, of the 1-isometry 4096x4096
#                                   sweep under both MFMA shapes, of the single image and the decoder; FETCH / WRITE traffic -> traffic.json
set -o pipefail
PART=${1:-A}; TAG=${2:-r03F}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
if [ "$PART" = A ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/${TAG}_pytest.log; tail -4 $O/${TAG}_pytest.log
  [ $rc -eq 0 ] || exit $rc
  timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
  echo "bench default ok"
  timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  timeout -k 10 400 python tools/q_stats.py 512,8,8,64 512,8,8,64,U,5 512,8,8,64,S 512,8,1,64 512,8,1,64,S 512,8,8,1 512,8,1,1 \
    2048,4,1,1 2048,4,1,1,U,2 4096,8,8,1 4096,8,1,1 1024,8,8,24 512,4,8,64 4096,16,1,1 4096,16,8,1 2048,8,1,4 > $O/${TAG}_sweep_matrix.txt 2>&1 || { tail $O/${TAG}_sweep_matrix.txt; exit 1; }
  FIC_Q_SHAPE=2 timeout -k 10 200 python tools/q_stats.py 4096,8,1,1 4096,16,1,1 >> $O/${TAG}_sweep_matrix.txt 2>&1
  grep "^W=" $O/${TAG}_sweep_matrix.txt | cut -c1-260
  timeout -k 10 300 python tools/oneshot_timing.py > $O/${TAG}_oneshot.json 2> $O/${TAG}_oneshot.err || { tail $O/${TAG}_oneshot.err; exit 1; }
  timeout -k 10 200 python tools/decode_timing.py > $O/${TAG}_decode_timing.json 2> $O/${TAG}_decode.err || { tail $O/${TAG}_decode.err; exit 1; }
  FIC_FAKE_DEVICES=8 timeout -k 10 300 python bench.py --inproc --gpus 4 --workload cfg4 --size 2048 --steps 3 --warmup 1 > $O/${TAG}_inproc4_fake.json 2> $O/${TAG}_inproc.err || { tail $O/${TAG}_inproc.err; exit 1; }
  timeout -k 10 300 python bench.py --inproc --gpus 1 --workload cfg4 --size 2048 --steps 3 --warmup 1 > $O/${TAG}_inproc1.json 2>> $O/${TAG}_inproc.err || { tail $O/${TAG}_inproc.err; exit 1; }
  FIC_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --extra-size 2048 > $O/${TAG}_bench_gloo2.json 2> $O/${TAG}_bench_gloo2.err || { tail -20 $O/${TAG}_bench_gloo2.err; exit 1; }
  echo "part A ok"
  exit 0
fi
if [ "$PART" = D ]; then       # after a late change: the bench line, the single image, the traffic stamp of the final csrc/
  timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { tail -20 $O/${TAG}_bench_default.err; exit 1; }
  export TMPDIR=/tmp; cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_single8 -- python3 $R/tools/single_image_trace.py 8 > $O/${TAG}_single8_under_rocprof.json 2> $O/${TAG}_prof_single8.err || exit 1
  cp "$(find $O/${TAG}_prof_single8 -name '*kernel_stats.csv' | head -1)" $O/${TAG}_single8_kernel_stats.csv && rm -rf $O/${TAG}_prof_single8
  cd $R && bash tools/gpu_traffic.sh ${TAG} --no-verify
  exit 0
fi
if [ "$PART" = C ]; then
  timeout -k 10 300 python tools/rgb_timing.py > $O/${TAG}_rgb.json 2> $O/${TAG}_rgb.err || { tail $O/${TAG}_rgb.err; exit 1; }
  FIC_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 4 --steps 5 --extra-size 2048 > $O/${TAG}_bench_gloo4.json 2> $O/${TAG}_bench_gloo4.err || { tail -20 $O/${TAG}_bench_gloo4.err; exit 1; }
  bash tools/gpu_fuzz_soak.sh ${TAG} 40 301 302 303 304 305 306 || exit 1
  echo "part C ok"
  exit 0
fi
export TMPDIR=/tmp; cd /tmp
stats() {   # stats <name> <command...>: rocprofv3 kernel stats -> $O/${TAG}_<name>_kernel_stats.csv
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$name -- "$@" > $O/${TAG}_${name}_under_rocprof.json 2> $O/${TAG}_prof_$name.err || { tail -20 $O/${TAG}_prof_$name.err; return 1; }
  cp "$(find $O/${TAG}_prof_$name -name '*kernel_stats.csv' | head -1)" $O/${TAG}_${name}_kernel_stats.csv && rm -rf $O/${TAG}_prof_$name
}
stats cfg2 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-verify || exit 1
for w in cfg3 cfg4 cfg4iso1 cfg5; do stats $w python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-verify || exit 1; done
stats single8 python3 $R/tools/single_image_trace.py 8 || exit 1
stats single1 python3 $R/tools/single_image_trace.py 1 || exit 1
stats decode python3 $R/tools/decode_timing.py 4096_S || exit 1
pmc() {     # pmc <name> <counters> <command...>
  local name=$1 ctr=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/${TAG}_pmc_$name -- "$@" > $O/${TAG}_pmc_$name.json 2> $O/${TAG}_pmc_$name.err || { tail -5 $O/${TAG}_pmc_$name.err; return 1; }
}
B2="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-verify"
pmc cfg2_a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" $B2 && \
pmc cfg2_b "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $B2 && \
pmc cfg2_e "TCC_HIT_sum TCC_MISS_sum" $B2 || exit 1
# the reference algorithm at 4096x4096: k_sweep_q16 (16x16x32, the default there) against k_sweep_q<4, 0> (32x32x16, FIC_Q_SHAPE=2)
pmc cfg4iso1_q16_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $B2 --workload cfg4iso1 || exit 1
export FIC_Q_SHAPE=2
pmc cfg4iso1_q32_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $B2 --workload cfg4iso1 || exit 1
unset FIC_Q_SHAPE
pmc cfg3_b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $B2 --workload cfg3 || exit 1
python3 - <<PY > $O/${TAG}_pmc_summary.txt
import csv, glob, collections
print("rocprofv3 --pmc passes (separate runs, --kernel-trace only) over: bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-verify [--workload ...]; per-launch averages")
for name in ["cfg2_a", "cfg2_b", "cfg2_e", "cfg4iso1_q16_b", "cfg4iso1_q32_b", "cfg3_b"]:
    for f in glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"]
            if "k_sweep" in kn or (name.startswith("cfg2") and ("k_pool_q" in kn or "k_range_q" in kn)):
                k = (kn.split("(")[0][:40], row["Counter_Name"]); acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v / n), "launches=%d" % n)
PY
cat $O/${TAG}_pmc_summary.txt
rm -rf $O/${TAG}_pmc_cfg*
cd $R && bash tools/gpu_traffic.sh ${TAG} --no-verify
