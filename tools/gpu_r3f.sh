#!/bin/bash
# full GPU suite + decoder kernel trace
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -o dec -- python3 tools/decode_timing.py 4096_S > gpurun_out/r03f_dec_prof.log 2>&1
echo "prof rc=$?"
find gpurun_out/prof_dec -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03f_decode_kernel_stats.csv
head -12 gpurun_out/r03f_decode_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/prof_dec
python -m pytest tests -x -q -m gpu > gpurun_out/r03f_pytest_full.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03f_pytest_full.log
