cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in 1 3; do
  echo "== round $r FIC_Q_G3=$v (phases build)"
  FIC_HIP_SO=$GRAFT_REPO_ROOT/libfic_g3_${v}_ab.so python tools/phases.py 2>&1 | grep n_iso
done
for so in libfic_g3_1n_ab.so ""; do
  echo "== round $r so=${so:-default(G3=3)}"
  export FIC_HIP_SO=${so:+$GRAFT_REPO_ROOT/$so}
  python tools/single_image_trace.py 8 U; python tools/single_image_trace.py 1 U; python tools/single_image_trace.py 8 N
  timeout -k 10 280 python tools/q_stats.py 2048,4,1,1 4096,8,1,1 4096,8,8,1 2048,8,8,1 1024,8,8,1 1024,8,1,1 2>&1 | grep "^W=" | cut -c1-250
  unset FIC_HIP_SO
done
done
