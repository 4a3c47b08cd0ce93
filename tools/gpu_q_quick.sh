#!/bin/bash
# Quick k_sweep_q timing matrix (stats only, no PMC).  Usage: tools/gpu_q_quick.sh <tag> [case ...]   case = W,B,iso,planes[,dist[,sweep[,chunks]]]
set -o pipefail
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 600 python tools/q_stats.py "$@" > $O/${TAG}_stats.txt 2>&1; rc=$?
cat $O/${TAG}_stats.txt
exit $rc
