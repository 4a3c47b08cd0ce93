"""GPU-side checks of the multi-rank plumbing and of the runtimes the library runs on:
  * the nccl (= RCCL) backend path of sharding.gather_records on CUDA tensors (world size 1 on a one-GPU box -- the same
    calls the 8-GPU run makes),
  * bench.py end to end: bare `--gpus 1`, and `--gpus 2` self-launched (gloo rehearsal: two ranks share the one GPU),
  * libfic_hip.so in a process that never imports torch (ctypes only): the ROCm runtime of /opt/rocm, which is what a JNI
    host gets, against the torch-bundled runtime the rest of the suite runs on -- same bits, timings recorded side by side."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import same_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(extra)
    return e


NCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import fic_amd
from fic_amd import synth
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
g = synth.image_s(256, 256, 11)
enc = fic_amd.ShardedEncoder(256, 256, 8, None, 8, planes=1, device=0)
enc.set_gray(torch.from_numpy(g.copy()).cuda().view(1, 256, 256))
enc.encode_local(torch.cuda.current_stream())
got = enc.gather()                                    # device-side records + dist.gather on the nccl backend
rec = enc.enc.records_device()[:, :, :].contiguous()
full = fic_amd.gather_records(rec, enc.spans, None, 0)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier(device_ids=[0])
want = enc.enc.results()
ok = (got["idx_local"] == want["idx_local"]).all() and (got["iso"] == want["iso"]).all() and (got["qrows"] == want["qrows"]).all()
ok = ok and (got["a"].view(np.uint32) == want["a"].view(np.uint32)).all() and (got["b"].view(np.uint32) == want["b"].view(np.uint32)).all()
back = fic_amd.unpack_records(full.cpu().numpy())
ok = ok and (back["idx_local"] == want["idx_local"]).all() and (back["qrows"] == want["qrows"]).all()
enc.close()
dist.destroy_process_group()
print("NCCL_OK" if ok and dist.is_nccl_available() else "NCCL_BAD")
"""


def test_nccl_backend_gather_on_cuda_tensors():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-c", NCCL_WORLD1.format(root=ROOT, port=port)], capture_output=True, text=True,
                       env=_clean_env(), timeout=600)
    assert r.returncode == 0 and "NCCL_OK" in r.stdout, r.stdout + r.stderr


def _bench(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       env=_clean_env(**(env or {})), timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_bare_single_gpu_line():
    j = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--planes", "2", "--cpu-budget", "1"])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "range-block matches/s"
    assert j["roofline"]["bound"] in ("mfma", "valu") and 0 < j["roofline"]["frac"] < 1.5
    assert j["roofline_hbm_logical"]["frac"] > 0 and j["cpu_baseline"]["value"] > 0
    assert j["valu_only"]["value"] > 0 and j["config"]["sweep_kind"] in (3, 6)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` from a bare shell (what the driver runs on the 8-GPU node): two fresh rank processes,
    here sharing the one GPU with records travelling through gloo; weak and strong scaling."""
    j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--planes", "2"], {"FIC_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and len(j["per_rank_sweep_ms"]) == 2 and j["value"] > 0
    j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg4", "--size", "512", "--scaling", "strong"],
               {"FIC_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0


def test_bench_default_line_carries_every_block():
    """The default line (what the driver runs at N = 1): the headline plus `verified`, `sustained`, `other_configs`
    (BASELINE configs 3, 4, 4 with 1 isometry, 5) and `strong_cfg4`.  The big images are shrunk for the test."""
    j = _bench(["--steps", "2", "--warmup", "1", "--cpu-budget", "1", "--sustain", "0.3", "--extra-size", "512", "--extra-steps", "2"])
    assert j["verified"] is True and j["verified_detail"]["against"] == "k_sweep_d4"
    assert j["config"]["sweep_kind"] == 6 and j["roofline"]["kernel"] == "k_sweep_q<4, 2, false>"
    assert 0 < j["roofline"]["executed_frac"] <= j["roofline"]["frac"]
    assert j["roofline_hbm_logical"]["ranges_per_pool_read"] == 32
    assert j["sustained"]["seconds"] >= 0.3 and j["sustained"]["value"] > 0 and j["sustained"]["clock_ghz"] > 0.5
    assert set(j["other_configs"]) == {"cfg3", "cfg4", "cfg4iso1", "cfg5"}
    for name, o in j["other_configs"].items():
        assert o["verified"] is True and o["value"] > 0 and o["ms_per_encode"] > 0 and o["roofline"]["frac"] > 0, name
    assert j["strong_cfg4"]["n_gpus"] == 1 and j["strong_cfg4"]["value"] == j["other_configs"]["cfg4"]["value"]
    assert j["single_image"]["ms"] > 0 and j["valu_only"]["value"] > 0 and j["pipelined"]["value"] > 0 and j["cpu_baseline"]["value"] > 0


def test_bench_two_ranks_default_line_has_the_strong_scaling_block():
    """`python bench.py --gpus 2` with no workload flags (what the driver runs on the multi-GPU node): weak cfg2 headline and
    the strong-scaling block, config 4's range blocks sharded over the two ranks through the same gather."""
    j = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--extra-size", "512", "--extra-steps", "2"], {"FIC_BENCH_BACKEND": "gloo"})
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["verified"] is True
    s4 = j["strong_cfg4"]
    assert s4["n_gpus"] == 2 and s4["scaling"] == "strong" and s4["verified"] is True and s4["value"] > 0
    assert s4["ranges_per_rank"] * 2 == s4["N_r"] and len(s4["per_rank_sweep_ms"]) == 2
    assert "other_configs" not in j


NO_TORCH = r"""
import sys, time, json
sys.path.insert(0, {root!r})
import numpy as np
import fic_amd
from fic_amd import capi, synth
assert "torch" not in sys.modules
g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
out = {{}}
for n_iso in (1, 8):
    r = capi.encode_gray_oneshot(g, 8, None, n_iso)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        capi.encode_gray_oneshot(g, 8, None, n_iso)
        ts.append(time.perf_counter() - t0)
    out[str(n_iso)] = {{"ms": sorted(ts)[len(ts) // 2] * 1e3, "idx": r["idx_local"].tolist(), "iso": r["iso"].tolist(),
                       "a": r["a"].view(np.uint32).tolist(), "b": r["b"].view(np.uint32).tolist()}}
assert "torch" not in sys.modules
import ctypes
out["hip_runtime"] = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1]
json.dump(out, open({dst!r}, "w"))
"""


def test_library_without_torch_in_the_process(tmp_path):
    """ctypes only, torch never imported: libfic_hip.so runs on /opt/rocm's libamdhip64 -- the runtime of the JNI / C++
    deployment -- and returns the same bits as in this (torch-runtime) process; both one-shot timings are recorded."""
    import time
    from fic_amd import capi
    dst = str(tmp_path / "no_torch.json")
    r = subprocess.run([sys.executable, "-c", NO_TORCH.format(root=ROOT, dst=dst)], capture_output=True, text=True,
                       env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    other = json.load(open(dst))
    g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
    rec = {"workload": "one 512x512 U image, B=8, full search, one-shot fic_encode_gray_u8 (host buffers in/out), median of 20",
           "hip_runtime_no_torch": other["hip_runtime"],
           "hip_runtime_here": [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1]}
    for n_iso in (1, 8):
        mine = capi.encode_gray_oneshot(g, 8, None, n_iso)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            capi.encode_gray_oneshot(g, 8, None, n_iso)
            ts.append(time.perf_counter() - t0)
        o = other[str(n_iso)]
        assert (mine["idx_local"] == np.array(o["idx"], np.int32)).all() and (mine["iso"] == np.array(o["iso"], np.int32)).all()
        assert (mine["a"].view(np.uint32) == np.array(o["a"], np.uint32)).all()
        assert (mine["b"].view(np.uint32) == np.array(o["b"], np.uint32)).all()
        rec[f"n_iso={n_iso}"] = {"ms_no_torch_process": o["ms"], "ms_torch_runtime_process": sorted(ts)[len(ts) // 2] * 1e3}
    assert rec["hip_runtime_no_torch"] and "/opt/rocm" in os.path.realpath(rec["hip_runtime_no_torch"][0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "runtime_compare.json"), "w"), indent=1)
    capi.release_cache()
