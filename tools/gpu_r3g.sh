#!/bin/bash
# round 3 session g: new prep kernels (k_pool_q<B>, k_range_q8) parity + kernel stats; decoder kernel stats; MFMA shape microbench
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests/test_gpu_q.py tests/test_gpu_bench_geometry.py tests/test_gpu_parity.py tests/test_gpu_d4.py -x -q -m gpu > $O/r03g_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/r03g_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03g_prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-verify > $O/r03g_prof_bench.json 2> $O/r03g_prof.err || { tail -20 $O/r03g_prof.err; exit 1; }
f=$(find $O/r03g_prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03g_cfg2_kernel_stats.csv; rm -rf $O/r03g_prof
cut -c1-150 $O/r03g_cfg2_kernel_stats.csv | head -8
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03g_profd -- python3 tools/decode_timing.py 4096_S > $O/r03g_dec.json 2> $O/r03g_dec.err || { tail -20 $O/r03g_dec.err; exit 1; }
f=$(find $O/r03g_profd -name "*kernel_stats.csv" | head -1); cp "$f" $O/r03g_decode_kernel_stats.csv; rm -rf $O/r03g_profd
cat $O/r03g_dec.json; cut -c1-150 $O/r03g_decode_kernel_stats.csv | head -12
tools/bin/mfma_shape 40 > $O/r03g_mfma_shape.txt 2>&1; cat $O/r03g_mfma_shape.txt
FIC_Q_NOFLAG=1 python tools/q_stats.py 512,8,8,1 512,8,1,1 2>&1 | grep "^W=" | cut -c1-200 > $O/r03g_single_floor.txt
python tools/q_stats.py 512,8,8,1 512,8,1,1 2>&1 | grep "^W=" | cut -c1-200 >> $O/r03g_single_floor.txt; cat $O/r03g_single_floor.txt
