#!/bin/bash
# round 3 session i: forced-shape tests, the default bench line, rocprof stats of it
mkdir -p gpurun_out
O=gpurun_out
python -m pytest tests/test_gpu_q.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/r03i_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/r03i_pytest.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 2 > $O/r03i_bench.json 2> $O/r03i_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open("gpurun_out/r03i_bench.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], "verified", j["verified"], "roofline", {k:j["roofline"][k] for k in ("frac","executed_frac","avg_launch_ms","clock_ghz","kernel")})
print("sustained", {k:j["sustained"][k] for k in ("value","ms_per_step","avg_launch_ms","clock_ghz")})
for k,v in (j.get("other_configs") or {}).items(): print(k, round(v["ms_per_encode"],3), v["kernel"], v["roofline"]["frac"], v["roofline"]["executed_frac"], v["verified"])
print("single", {k:j["single_image"][k] for k in ("ms","latency_ms","sweep_ms","kernel")})
print("valu", j["valu_only"]["value"], "pipelined", j["pipelined"]["value"], "cpu", j["cpu_baseline"]["value"])
PY
