# same-box A/B of the prefix seed: the default library against a build without it.  Build the variant first (in the container):
#   cd fractal-image-compression_amd && FIC_HIPCC_FLAGS=-DFIC_Q_SEED=0 python build.py && cp libfic_hip.so ../libfic_noseed_ab.so && python build.py
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for so in "" libfic_noseed_ab.so; do
  echo "== round $r so=${so:-default(seed)}"
  FIC_HIP_SO=${so:+$GRAFT_REPO_ROOT/$so} timeout -k 10 280 python tools/q_stats.py 4096,8,8,1 4096,8,1,1 2048,4,1,1 2048,8,8,1 1024,8,8,24 4096,16,8,1 2>&1 | grep "^W=" | cut -c1-120
done
done
