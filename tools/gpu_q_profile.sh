#!/bin/bash
# k_sweep_q diagnosis: stats counters + SQ PMC passes on the default bench workload.  Usage: tools/gpu_q_profile.sh <tag> [bench args]
set -o pipefail
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 300 python - "$@" <<'PY' > $O/${TAG}_stats.txt 2>&1 || { tail -20 $O/${TAG}_stats.txt; exit 1; }
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import fic_amd
from fic_amd import synth
def run(W, B, n_iso, planes, dist="U", sweep=6, chunks=0, reps=5):
    f = synth.image_u if dist == "U" else synth.image_s
    g = np.stack([f(W, W, 100 + 3 * p) for p in range(planes)])
    d = torch.from_numpy(g).cuda()
    enc = fic_amd.Encoder(W, W, B, None, n_iso, planes)
    enc.set_gray(d); enc.set_option("sweep", sweep); enc.set_option("time_sweep", 1)
    if chunks: enc.set_option("chunks", chunks)
    if sweep == 6: enc.set_option("sweep_stats", 1)
    s = torch.cuda.current_stream()
    for _ in range(2): enc.encode(0, -1, s)
    enc.sync(); enc.sweep_time()
    if sweep == 6: enc.sweep_stats()
    for _ in range(reps): enc.encode(0, -1, s)
    enc.sync()
    ms, n = enc.sweep_time()
    st = enc.sweep_stats() if sweep == 6 else {}
    info = enc.info()
    evals = planes * enc.n_ranges * enc.n_domains * n_iso
    nk = B * B // 16
    tiles = st.get("tiles", 0) / reps if st else 0
    print(f"W={W} B={B} iso={n_iso} planes={planes} dist={dist} sweep={sweep} chunks={info['chunks']}: {ms/n:.3f} ms/sweep, "
          f"{evals/(ms/n*1e-3):.3e} evals/s, mfma_frac={evals*2*B*B/(ms/n*1e-3)/2.5e15:.3f}"
          + (f", cycles/tile@2.4GHz={ms/n*1e-3*2.4e9*1024/max(tiles,1):.0f} (floor {32*nk}), flagged_tiles={st['flagged_tiles']/max(st['tiles'],1):.4f}, "
             f"exact_pairs/range={st['exact_pairs']/reps/(planes*enc.n_ranges):.1f}, waves={st['waves']//reps}" if st else ""), flush=True)
    enc.close()
run(512, 8, 8, 64)
run(512, 8, 8, 64, dist="S")
run(512, 8, 8, 64, sweep=3)
run(512, 8, 1, 64)
run(512, 8, 1, 64, sweep=3)
run(512, 8, 8, 1)
run(512, 8, 8, 1, chunks=8)
run(512, 8, 8, 1, chunks=16)
run(512, 8, 8, 1, sweep=3)
run(2048, 4, 1, 1)
run(2048, 4, 1, 1, sweep=3)
run(1024, 4, 8, 4)
run(1024, 4, 8, 4, sweep=3)
run(2048, 16, 1, 1)
run(2048, 16, 1, 1, sweep=3)
run(2048, 16, 8, 1)
run(2048, 16, 8, 1, sweep=3)
run(4096, 8, 8, 1, reps=2)
PY
cat $O/${TAG}_stats.txt
export TMPDIR=/tmp; cd /tmp
run_pass() {
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/${TAG}_pmc_$1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/${TAG}_pmc_$1.json 2> $O/${TAG}_pmc_$1.err || { tail -5 $O/${TAG}_pmc_$1.err; return 1; }
}
run_pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" && \
run_pass b "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" && \
run_pass c "SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES_EQ_64"
python3 - <<PY
import csv, glob, collections
for name in ["a","b","c"]:
    for f in glob.glob("$O/${TAG}_pmc_%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for row in csv.DictReader(open(f)):
            if "k_sweep" in row["Kernel_Name"]:
                k=(row["Kernel_Name"][:34], row["Counter_Name"]); acc[k][0]+=float(row["Counter_Value"]); acc[k][1]+=1
        for k,(v,n) in sorted(acc.items()):
            print(name, k[0], k[1], "avg/launch=%.6g" % (v/n))
PY
