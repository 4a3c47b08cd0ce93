#!/bin/bash
# Same-box A/B of whole source revisions of csrc/ (snapshots under _ab/<name>/, made with `git archive <rev> .../csrc`):
# builds each snapshot and the working tree into libfic_hip.so in turn and times the same cases.  The working tree is built last.
# Usage: tools/gpu_ab_rev.sh <tag> "<dir1> <dir2> ..." case...
set -o pipefail
TAG=$1; DIRS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
P=fractal-image-compression_amd
build() {  # $1 = csrc dir
  local src=""; for f in fic_prep.hip fic_sweep.hip fic_d4.hip fic_mfma.hip fic_bf16.hip fic_q.hip fic_rgb.hip fic_decode.hip fic_capi.cpp fic_capi_decode.cpp fic_capi_rgb.cpp fic_capi_multi.cpp; do [ -f $1/$f ] && src="$src $1/$f"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -I$R/include -I$1 $src -o $P/libfic_hip.so 2> $O/${TAG}_build.err || { tail -5 $O/${TAG}_build.err; return 1; }
}
for round in 1 2; do
for d in $DIRS $P/csrc; do
  build $d || exit 1
  echo "== round $round rev=$d" | tee -a $O/${TAG}_ab.txt
  timeout -k 10 300 python tools/q_stats.py "$@" 2>&1 | grep "^W=" | sed 's/ms\/sweep.*mfma_frac/ms mfma_frac/; s/, cycles[^,]*(floor [0-9]*)//' | tee -a $O/${TAG}_ab.txt
done
done
