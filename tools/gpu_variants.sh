#!/bin/bash
# A/B of compile-time variants of k_sweep_q on ONE box (CDNA guide rule 24): builds each variant on the box, times the same cases.
# Usage: tools/gpu_variants.sh <tag> "<flags1>|<flags2>|..." case...
set -o pipefail
TAG=$1; VARS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
IFS='|' read -ra V <<< "$VARS"
for round in 1 2; do
for v in "${V[@]}"; do
  FIC_HIPCC_FLAGS="$v" python -c "import importlib.util,sys; s=importlib.util.spec_from_file_location('b','fractal-image-compression_amd/build.py'); m=importlib.util.module_from_spec(s); s.loader.exec_module(m); m.build(force=True)" > /dev/null 2> $O/${TAG}_build.err || { tail -5 $O/${TAG}_build.err; exit 1; }
  echo "== round $round flags='$v'" | tee -a $O/${TAG}_variants.txt
  timeout -k 10 300 python tools/q_stats.py "$@" 2>&1 | grep "^W=" | sed 's/, cycles.*//' | tee -a $O/${TAG}_variants.txt
done
done
