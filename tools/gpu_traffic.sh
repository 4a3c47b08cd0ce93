#!/bin/bash
# HBM-side traffic of the sweep kernel of the default bench command: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (MI355X_MICROARCH.md "HBM": TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2), per launch, with the gfx950 correction
# (FETCH_SIZE reports half the bytes of wide coalesced reads -> x2).  Writes profiles/traffic.json keyed by workload + kernel
# and stamped with the hash of csrc/ that bench.py checks.   Usage: tools/gpu_traffic.sh <tag> [bench args]
set -o pipefail
TAG=${1:-traffic}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt "$@" > $O/${TAG}_pmc_$c.json 2> $O/${TAG}_pmc_$c.err || { tail -5 $O/${TAG}_pmc_$c.err; exit 1; }
done
python3 - "$O" "$TAG" "$R" <<'PY'
import csv, glob, json, os, sys, collections
O, TAG, R = sys.argv[1:4]
sys.path.insert(0, R)
import bench
line = json.loads([l for l in open(f"{O}/{TAG}_pmc_FETCH_SIZE.json") if l.startswith("{")][0])
kname = line["roofline"]["kernel"]
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = [0.0, 0]
    for f in glob.glob(f"{O}/{TAG}_pmc_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if kname.split("<")[0] in row["Kernel_Name"] and row["Counter_Name"] == c:
                acc[0] += float(row["Counter_Value"]); acc[1] += 1
    vals[c] = acc[0] / max(acc[1], 1)
    print(c, "avg KB/launch = %.6g over %d launches" % (vals[c], acc[1]))
cfg = line["config"]
planes = cfg.get("planes_per_rank", cfg.get("planes"))
wl = cfg["workload"].split(":")[0]
key = f"{wl}:planes={planes}:n_iso={cfg['n_iso']}:kernel={kname}:gpus={line['n_gpus']}"
ent = {"bytes_per_launch": int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024), "csrc_hash": bench.csrc_hash(),
       "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
       "how": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3` ({kname}): FETCH_SIZE "
              f"{vals['FETCH_SIZE']:.0f} KB x 2 (gfx950 halves wide reads: 16-byte fragment loads) + WRITE_SIZE {vals['WRITE_SIZE']:.0f} KB, per launch"}
path = os.path.join(R, "gpurun_out", "traffic.json")
try:
    tj = json.load(open(os.path.join(R, "profiles", "traffic.json")))
except Exception:
    tj = {}
tj[key] = ent
json.dump(tj, open(path, "w"), indent=1)
print(key, ent)
PY
