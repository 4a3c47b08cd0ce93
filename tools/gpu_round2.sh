#!/bin/bash
# GPU-box session for the matrix-core comparison: bench lines for both sweeps on every configuration, rocprof
# kernel stats of the default bench and of --sweep 3, the MFMA/VALU co-issue microbenchmark.  Usage: tools/gpu_round2.sh <tag>
set -o pipefail
TAG=${1:-r01y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
line() { python3 -c "
import sys, json
d = json.loads(open(sys.argv[1]).read())
r = d['roofline']
print('%-44s value %.4g %s  ms/step %.4g  sweep %.4g ms  kernel %s  roofline(%s) frac %.3f' % (sys.argv[2], d['value'], d['unit'], d['ms_per_step'], r['avg_launch_ms'], r['kernel'], r['bound'], r['frac']))
" "$1" "$2"; }
for spec in "cfg2:10:2" "cfg2x1:20:5" "cfg4:2:1" "cfg4iso1:3:1" "cfg3:2:1" "cfg5:2:1"; do
  IFS=: read wl st wu <<< "$spec"
  for sw in 0 2 3; do
    f=$O/${TAG}_${wl}_s${sw}.json
    timeout -k 10 300 python bench.py --workload $wl --sweep $sw --steps $st --warmup $wu --no-cpu-baseline --no-alt > $f 2> $O/${TAG}_err.txt || { tail -5 $O/${TAG}_err.txt; exit 1; }
    line $f "$wl sweep=$sw" >> $O/${TAG}_matrix.txt
  done
done
for spec in "cfg4iso1:16:3:1" "cfg4:16:2:1" "cfg2:4:10:2"; do
  IFS=: read wl blk st wu <<< "$spec"
  for sw in 2 3; do
    f=$O/${TAG}_${wl}_B${blk}_s${sw}.json
    timeout -k 10 300 python bench.py --workload $wl --block $blk --sweep $sw --steps $st --warmup $wu --no-cpu-baseline --no-alt > $f 2> $O/${TAG}_err.txt || { tail -5 $O/${TAG}_err.txt; exit 1; }
    line $f "$wl B=$blk sweep=$sw" >> $O/${TAG}_matrix.txt
  done
done
cat $O/${TAG}_matrix.txt
hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-value tools/mfma_coissue.hip -o /tmp/mfma_coissue 2>/dev/null && timeout -k 10 120 /tmp/mfma_coissue > $O/${TAG}_mfma_coissue.txt 2>&1
cat $O/${TAG}_mfma_coissue.txt
export TMPDIR=/tmp
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_s3 -- python3 $R/bench.py --sweep 3 --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_prof_s3_bench.json 2> $O/${TAG}_prof.err || { tail -20 $O/${TAG}_prof.err; exit 1; }
find $O/${TAG}_prof_s3 -name "*kernel_stats*" | head -2
