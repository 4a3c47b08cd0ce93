# same-box A/B: k_sweep_qs (theta_g refreshed in the fast path of short pool chunks) against a build without it.  Build the variant first:
#   cd fractal-image-compression_amd && FIC_HIPCC_FLAGS=-DFIC_Q_GFAST_TILES=0 python build.py && cp libfic_hip.so ../libfic_nogfast_ab.so && python build.py
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for so in libfic_nogfast_ab.so ""; do
  echo "== round $r so=${so:-default}"
  export FIC_HIP_SO=${so:+$GRAFT_REPO_ROOT/$so}
  python tools/single_image_trace.py 8 U; python tools/single_image_trace.py 1 U; python tools/single_image_trace.py 8 N; python tools/single_image_trace.py 1 N
  timeout -k 10 280 python tools/q_stats.py 2048,4,1,1 4096,8,1,1 2048,8,8,1 1024,8,8,1 1024,8,1,1 256,8,8,1 256,8,8,16 2>&1 | grep "^W=" | cut -c1-200
  unset FIC_HIP_SO
done
done
