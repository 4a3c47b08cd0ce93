#!/bin/bash
# Fuzz soak: the seeded fuzz tests of the grey sweeps (every kernel, incl. k_sweep_q with random chunk counts) and of the RGB
# full-search sweeps against the oracle, several seeds x FIC_FUZZ_CASES cases.  Usage: tools/gpu_fuzz_soak.sh <tag> <cases> seed...
set -o pipefail
TAG=$1; CASES=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
: > $O/${TAG}_fuzz.txt
for s in "$@"; do
  for t in tests/test_gpu_parity.py::test_seeded_fuzz_geometries_windows_sweeps tests/test_gpu_rgb_q.py::test_seeded_fuzz_rgb_full_search; do
    FIC_FUZZ_SEED=$s FIC_FUZZ_CASES=$CASES timeout -k 10 500 python -m pytest $t -x -q > $O/${TAG}_fuzz_last.log 2>&1
    r=$(tail -1 $O/${TAG}_fuzz_last.log)
    echo "seed $s $t: $r" | tee -a $O/${TAG}_fuzz.txt
    case "$r" in *passed*) ;; *) tail -30 $O/${TAG}_fuzz_last.log | tee -a $O/${TAG}_fuzz.txt; exit 1;; esac
  done
done
