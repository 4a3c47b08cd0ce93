"""bench.py's launch contract on the CPU: `python bench.py --gpus N` from a bare shell starts its own N rank
processes (before anything touches the GPU), waits for them and propagates failure; under torch.distributed.run the
given environment is used.  `--probe` runs the launcher + rendezvous + codebook gather on gloo with synthetic records
(no GPU, no libfic_hip.so); its output is not a benchmark line."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bare_shell_self_launch(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--probe"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["probe"] is True and j["n_gpus"] == n and j["gather_ok"] is True
    assert j["rank_env"] == {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1"}


def test_failed_rank_fails_the_launch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--probe", "--probe-fail-rank", "1"], capture_output=True, text=True,
                       env=_env(), timeout=300)
    assert r.returncode != 0


def test_single_process_probe_needs_no_launcher():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--probe"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_under_torch_distributed_run_the_given_environment_is_used():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--probe"], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["gather_ok"] is True


def test_launch_ranks_environment_and_worst_exit_code():
    sys.path.insert(0, ROOT)
    import bench
    code = "import os,sys; r=int(os.environ['RANK']); assert os.environ['WORLD_SIZE']=='3' and os.environ['LOCAL_RANK']==str(r); sys.exit(5 if r==2 else 0)"
    assert bench.launch_ranks(3, [sys.executable, "-c", code], env=_env()) == 5
    assert bench.launch_ranks(2, [sys.executable, "-c", "import os; assert os.environ['MASTER_ADDR']=='127.0.0.1'"], env=_env()) == 0
    # a hung peer of a failed rank is stopped instead of waited for
    hang = "import os,sys,time; r=int(os.environ['RANK']); sys.exit(3) if r==0 else time.sleep(600)"
    assert bench.launch_ranks(2, [sys.executable, "-c", hang], env=_env()) == 3


def test_a_rank_that_never_reaches_the_rendezvous_ends_the_launch_inside_the_deadline():
    """ranks that hang in init_process_group / the first gather must not leave a silent job behind.  Rank 1
    sleeps instead of joining; FIC_BENCH_TIMEOUT bounds the launcher (and every rank's own watchdog)."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--probe", "--probe-hang-rank", "1"], capture_output=True, text=True,
                       env=dict(_env(), FIC_BENCH_TIMEOUT="10"), timeout=200)
    assert r.returncode != 0
    assert time.time() - t0 < 90
    assert "FIC_BENCH_TIMEOUT" in r.stderr


def test_launch_ranks_deadline_names_the_live_ranks(capfd):
    sys.path.insert(0, ROOT)
    import bench
    code = "import os,sys,time; time.sleep(600) if os.environ['RANK']=='1' else sys.exit(0)"
    t0 = time.time()
    assert bench.launch_ranks(2, [sys.executable, "-c", code], env=_env(), deadline_s=3.0) == 124
    assert time.time() - t0 < 30
    assert "ranks [1] of 2 still running" in capfd.readouterr().err
