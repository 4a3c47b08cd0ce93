#!/bin/bash
# round 3, session a: the new tests at the benchmarked geometries, the hardened multi-device entry, the new bench line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bench_geometry.py tests/test_gpu_multi.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r03a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a_pytest.log
tail -5 gpurun_out/r03a_pytest.log
timeout -k 10 600 python bench.py --steps 20 --warmup 2 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r03a_bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03a_bench.json"))
print({k:(v if not isinstance(v,(dict,list)) else '...') for k,v in j.items()})
print("sustained", j.get("sustained"))
for k,v in (j.get("other_configs") or {}).items(): print(k, v["ms_per_encode"], v["value"], v["roofline"], v["verified"])
print("single", j.get("single_image"))
PY
