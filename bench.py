#!/usr/bin/env python3
"""bench.py -- range-block matches/s of the MI355X grey encode hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

One "step" = one pass of the hot path (pool build + range prep + sweep + finalise + codebook
gather) over one batch of synthetic grey input that is already resident in HBM.  Default
workload = BASELINE.json configs[1], the configuration the metric is quoted on:
512x512 grey, 8x8 range / 16x16 domain (B=8), full search, 8 isometries -- as a batch of 64
images per GPU so a step is milliseconds, not launch latency (the literal single image is timed
too and reported as `single_image`).  At N>1 every rank encodes its own batch (weak scaling:
units = images, no data-path collective) and the codebooks are gathered to rank 0 with one RCCL
gather per step (the path's only exchange), overlapped with the next step's sweep.
`--workload cfg4 --scaling strong` shards ONE 4096x4096 image's range blocks across the ranks.

Launch: `python bench.py --gpus N` from a bare shell starts its own N rank processes (fresh
children, spawned before this process touches torch or the GPU; it waits for them -- at most
FIC_BENCH_TIMEOUT seconds, default 900 -- and exits with their worst code); under
`python -m torch.distributed.run ... bench.py --gpus N` the RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* environment is used as given.  Every rank carries a watchdog with the same deadline.

Prints ONE JSON line on rank 0 (contract in the task statement).  Beside the contract's keys:
  roofline              the compute bound of the sweep kernel that ran (matrix-core FLOP/s against the dense MFMA peak, or
                        VALU issue for the VALU-only sweeps): `frac` algorithmic, `executed_frac` what the matrix cores issue
  roofline_hbm_logical  SURVEY 8d's byte model, which the sweep exceeds by the register-tile reuse factor
  verified              the codebook of the timed configuration, re-encoded after the timed region by the VALU-only sweep
                        (another kernel family) and compared record by record on the device
  sustained             the same step loop for >= 2 s of wall time, with the clock the chip held during it
  other_configs         BASELINE.json configs 3, 4 (8 and 1 isometries), 5: a few steps each (N = 1)
  strong_cfg4           config 4 range-sharded over the N ranks through the same gather (every N): north_star's
                        strong-scaling point
  valu_only, pipelined, single_image, cpu_baseline (the C oracle, 1 core, bounded sample)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process GPU buffers need dmabuf IPC on this pool (already exported there; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: (W, H, B, n_iso, planes per rank (weak) / total (strong), default scaling, seed key, description)
    "cfg2": dict(W=512, H=512, B=8, n_iso=8, planes=64, scaling="weak", seed="cfg2",
                 desc="cfg2: 512x512 synthetic grey U, 8x8 range/16x16 domain, full search (wK=125), 8 iso, batch of 64 images per GPU"),
    "cfg2x1": dict(W=512, H=512, B=8, n_iso=8, planes=1, scaling="weak", seed="cfg2",
                   desc="cfg2 single image: 512x512 synthetic grey U, B=8, full search, 8 iso"),
    "cfg3": dict(W=2048, H=2048, B=4, n_iso=1, planes=1, scaling="strong", seed="cfg3",
                 desc="cfg3: 2048x2048 synthetic grey U, 4x4 range/8x8 domain, full search (wK=1021), 1 iso"),
    "cfg4": dict(W=4096, H=4096, B=8, n_iso=8, planes=1, scaling="strong", seed="cfg4",
                 desc="cfg4: 4096x4096 synthetic grey U, 8x8 range/16x16 domain, full search (wK=1021), 8 iso, range blocks sharded across ranks"),
    "cfg4iso1": dict(W=4096, H=4096, B=8, n_iso=1, planes=1, scaling="strong", seed="cfg4",
                     desc="cfg4 parity mode: 4096x4096 synthetic grey U, B=8, full search, 1 iso (the reference algorithm)"),
    "cfg5": dict(W=1024, H=1024, B=8, n_iso=8, planes=24, scaling="weak", seed="cfg5",
                 desc="cfg5: 1024x1024 grey planes (RGB channels encoded independently), B=8, full search, 8 iso, 24 planes (8 RGB images) per GPU"),
}
OTHER_CONFIGS = ["cfg3", "cfg4", "cfg4iso1", "cfg5"]          # timed after the headline at N = 1 (`other_configs`)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "i8": 5000.0}   # dense peaks, MI355X_MICROARCH.md "Matrix cores"
PEAK_CLOCK_GHZ = 2.4                     # the clock those peaks are quoted at
# Measured on MI355X (profiles/r01_valu_issue_rate_microbench.txt): v_dot4_u32_u8 -- like v_mad_i32_i24,
# v_cvt_*, v_cmp_* and any VALU op with an SGPR source -- issues one wave64 instruction per 4 cycles
# per SIMD (only plain v_fma_f32 / v_add_u32 reach 2 cycles).  Peak for the VALU sweeps' instruction mix:
VALU_WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 4.0   # wave-instructions/s, chip-wide, at the 2.4 GHz max clock

# sweep kind (fic_ctx_info) -> (kernel family, operand type or None)
SWEEP_KINDS = {1: ("k_sweep_generic", None), 2: ("k_sweep_fast", None), 5: ("k_sweep_d4", None),
               3: ("matrix-core (exact covariances)", "bf16"), 4: ("matrix-core (i8 operands)", "i8"),
               6: ("matrix-core (normalised f16 prune GEMM)", "f16")}


def csrc_hash():
    """Identity of the kernel sources: profiles/traffic.json entries are only trusted for the tree they were measured on."""
    d = os.path.join(ROOT, "fractal-image-compression_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:12]


def kernel_name(kind, B, n_iso, chunks=1):
    if kind == 3:
        bf16 = B <= 8 or n_iso == 8
        return ("k_sweep_bf16" if bf16 else "k_sweep_mfma") + ("" if n_iso == 8 else ("_1" if bf16 else "1")), ("bf16" if bf16 else "i8")
    if kind == 4:
        return "k_sweep_mfma" + ("" if n_iso == 8 else "1"), "i8"
    if kind == 6:     # k_sweep_q<NK, MODE, MULTI>: NK = n/16 MFMA steps, MODE 0 = 1 isometry, 1 = 8 isometries (B = 4), 2 = 8 isometries
        # folded; MULTI: the launch has more than one pool chunk.  1 isometry at B = 8 / 16: k_sweep_q16<NK, MULTI> (16x16x32 MFMA)
        multi = 'true' if chunks > 1 else 'false'     # (1 isometry on LARGE pools at B = 8 / 16 runs k_sweep_q16: Encoder.last_kernel() knows)
        return f"k_sweep_q<{B * B // 16}, {0 if n_iso == 1 else (1 if B == 4 else 2)}, {multi}>", "f16"
    return SWEEP_KINDS.get(kind, ("k_sweep_fast", None))[0], None


# ----------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` from a bare shell
# ----------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def bench_timeout_s():
    try:
        return float(os.environ.get("FIC_BENCH_TIMEOUT", "900"))
    except ValueError:
        return 900.0


def launch_ranks(n, cmd, env=None, poll_s=0.2, deadline_s=None):
    """Starts `cmd` n times as fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set (one process per GPU; rendezvous on 127.0.0.1), waits for all of them and returns the worst exit code.  If a
    rank fails, the others are stopped (their exact PIDs) instead of being left at a barrier; if the ranks are still
    running after `deadline_s` (FIC_BENCH_TIMEOUT, default 900 s) -- a rendezvous or a collective that never
    completes -- the launcher says which ranks were alive, stops them and returns 124.  The calling process must not
    have touched the GPU: nothing here imports torch or loads libfic_hip.so."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    deadline_s = bench_timeout_s() if deadline_s is None else deadline_s
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e))
    t_end = time.time() + deadline_s
    worst = 0
    live = set(range(n))

    def stop(which):
        for o in sorted(which):
            procs[o].terminate()
        t_kill = time.time() + 10.0
        left = set(which)
        while left and time.time() < t_kill:
            left = {r for r in left if procs[r].poll() is None}
            time.sleep(poll_s)
        for o in sorted(left):
            procs[o].kill()
            procs[o].wait()

    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
        if live and worst:                       # a failed rank leaves the others waiting at a collective
            stop(live)
            live = set()
        elif live and time.time() > t_end:
            print(f"bench.py launcher: deadline of {deadline_s:.0f} s passed with ranks {sorted(live)} of {n} still running "
                  f"(FIC_BENCH_TIMEOUT); stopping them", file=sys.stderr, flush=True)
            stop(live)
            live = set()
            worst = worst or 124
        elif live:
            time.sleep(poll_s)
    return worst


class Watchdog:
    """Per-rank deadline (FIC_BENCH_TIMEOUT): a rank that is still running when it passes reports the stage it was in and
    exits with code 124, so that whatever launched the ranks (launch_ranks above, torch.distributed.run) sees a failure
    instead of waiting for a collective that never completes."""

    def __init__(self, rank, seconds):
        self.stage, self.rank = "start", rank
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.t.start()

    def _fire(self):
        print(f"bench.py rank {self.rank}: still running after FIC_BENCH_TIMEOUT in stage '{self.stage}'; giving up", file=sys.stderr, flush=True)
        os._exit(124)

    def cancel(self):
        self.t.cancel()


# ----------------------------------------------------------------------------------------------------------------
def cpu_baseline(wl, img, budget_s=12.0):
    """The oracle (C restatement of the Java loops, -O2, 1 thread) on a bounded sample of the
    same workload: the first ranges of image 0 against its full pool."""
    from oracle import fic_oracle as fo
    W, H, B, n_iso = wl["W"], wl["H"], wl["B"], wl["n_iso"]
    Rw, Rh, Dw, Dh = fo.geometry(W, H, B)
    argb = fo.gray_to_argb(img)
    t0 = time.perf_counter()
    fo.encode_gray(argb, W, H, B, Dw, n_iso, 0, 2)          # includes the pool build
    t_probe = time.perf_counter() - t0
    t0 = time.perf_counter()
    fo.encode_gray(argb, W, H, B, Dw, n_iso, 0, 4)
    t4 = time.perf_counter() - t0
    per_range = max((t4 - t_probe) / 2.0, 1e-6)
    n = int(max(4, min(Rw * Rh, budget_s / per_range)))
    t0 = time.perf_counter()
    fo.encode_gray(argb, W, H, B, Dw, n_iso, 0, n)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "range-block matches/s", "cores": 1, "kind": "port",
           "sample": f"first {n} of {Rw * Rh} range blocks of image 0 against the full {Dw * Dh}-block pool "
                     f"(x{n_iso} iso), pool build included, {dt:.1f} s; C restatement of the Java loops "
                     f"(oracle/fic_oracle.c, gcc -O2, no JVM in this image)"}
    # optional all-core figure (the reference itself is single-threaded): the same loops over disjoint range
    # slices, one thread per core (ctypes releases the GIL), about half the 1-core budget of wall time
    try:
        from concurrent.futures import ThreadPoolExecutor
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)          # a 1-GPU box's CPU share on this pool
        per = max(2, min(Rw * Rh // cores, int(0.5 * budget_s / per_range)))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda i: fo.encode_gray(argb, W, H, B, Dw, n_iso, i * per, (i + 1) * per), range(cores)))
        dta = time.perf_counter() - t0
        out["all_cores"] = {"value": cores * per / dta, "cores": cores, "sample": f"{cores} threads x {per} range blocks, {dta:.1f} s"}
    except Exception as e:  # never let the optional figure break the bench line
        out["all_cores"] = {"error": str(e)}
    return out


def probe(args, world, rank):
    """--probe: the launcher + rendezvous + codebook gather with synthetic records on the CPU (gloo), no GPU and no
    libfic_hip.so.  Used by tests/test_bench_launcher.py; its output is not a benchmark line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import fic_amd
    if args.probe_hang_rank is not None and rank == args.probe_hang_rank:
        time.sleep(3600)                       # a rank that never reaches the rendezvous
    if world > 1:
        from datetime import timedelta
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    spans = fic_amd.shard_spans(1000, 64, world)
    b, c = spans[rank]
    rec = torch.from_numpy(np.arange(b * 6, (b + c) * 6, dtype=np.int32).reshape(1, c, 6))
    full = fic_amd.gather_records(rec, spans, None, 0) if world > 1 else rec
    ok = None
    if rank == 0:
        ok = bool((full.numpy().reshape(-1) == np.arange(6000, dtype=np.int32)).all())
        print(json.dumps({"probe": True, "n_gpus": world, "gather_ok": ok, "spans": spans,
                          "rank_env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if args.probe_fail_rank is not None and rank == args.probe_fail_rank:
        sys.exit(7)
    return 0 if (ok or rank != 0) else 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"])
    ap.add_argument("--planes", type=int, default=None, help="override images per GPU (weak) / total (strong)")
    ap.add_argument("--n-iso", type=int, default=None, choices=[1, 8])
    ap.add_argument("--size", type=int, default=None, help="override image side (exploratory)")
    ap.add_argument("--block", type=int, default=None, choices=[4, 8, 16], help="override B (exploratory)")
    ap.add_argument("--dist", default="U", choices=["U", "S", "lena"],
                    help="synthetic input: U = iid bytes (the quoted distribution), S = flat tiles + noise, "
                         "lena = LenaGrey.png tiled/cropped to size with a per-image shift (robustness checks)")
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--sweep", type=int, default=0, choices=[0, 2, 3, 4, 5, 6],
                    help="0 = library default (matrix-core sweep for full search); 2 / 5 = VALU-only sweeps (k_sweep_fast / "
                         "k_sweep_d4, north_star's literal design); 3 / 4 = matrix-core sweeps with exact covariances "
                         "(bf16 / i8 operands); 6 = matrix-core sweep with the normalised f16 prune GEMM")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2],
                    help="contexts/streams that take the steps alternately (2: prep of step k+1 overlaps the tail of sweep k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timings (valu_only, pipelined, single_image, sustained, other_configs)")
    ap.add_argument("--no-extra", action="store_true", help="skip other_configs / strong_cfg4 only")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-run codebook check")
    ap.add_argument("--sustain", type=float, default=2.0, help="seconds of the `sustained` block (0 = skip)")
    ap.add_argument("--extra-steps", type=int, default=3, help="timed steps of every other_configs / strong_cfg4 entry")
    ap.add_argument("--extra-size", type=int, default=None, help=argparse.SUPPRESS)   # tests: shrink cfg3/cfg4 images
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--inproc", action="store_true",
                    help="time the in-library multi-device entry fic_encode_gray_u8_multi (one process, --gpus devices) instead")
    ap.add_argument("--probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-fail-rank", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--probe-hang-rank", type=int, default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
class Env:
    """What every Run needs to know about this rank."""

    def __init__(self, args, world, rank, local_rank, backend, dist, fic_amd, torch, np, watchdog):
        self.args, self.world, self.rank, self.local_rank, self.backend, self.dist = args, world, rank, local_rank, backend, dist
        self.fic_amd, self.torch, self.np, self.watchdog = fic_amd, torch, np, watchdog

    def barrier(self):
        if self.dist:
            if self.backend == "nccl":
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()

    def max_over_ranks(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, x):
        if not self.dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return float(t.item())

    def gather_floats(self, x):
        if not self.dist:
            return [x]
        mine = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        allv = [self.torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(allv, mine)
        return [float(v.item()) for v in allv]


class Run:
    """One workload on this rank: the synthetic batch resident in HBM, `ncontexts` encoder contexts with a compute stream
    each, and the step loop (encode + codebook gather).  Used for the headline and for every extra block, so that every
    number in the line comes from the same code."""

    def __init__(self, env, wl, scaling, ncontexts=1, sweep=0, chunks=0, dist_kind="U"):
        torch, fic_amd, np = env.torch, env.fic_amd, env.np
        self.env, self.wl, self.scaling, self.sweep = env, wl, scaling, sweep
        W, H, B, n_iso, planes = wl["W"], wl["H"], wl["B"], wl["n_iso"], wl["planes"]
        self.seed = fic_amd.synth.SEEDS[wl["seed"]]
        first = env.rank * planes if scaling == "weak" else 0
        seeds = [self.seed + 3 * (first + p) for p in range(planes)]
        self.dist_kind = dist_kind
        if dist_kind == "U":
            # synthetic input, generated on the device: resident in HBM before any timing (same integers as synth.image_u)
            self.dev_in = fic_amd.synth.images_u_torch(W, H, seeds, torch.device("cuda", env.local_rank))
        else:
            self.dev_in = torch.from_numpy(np.stack([self.make_image(s) for s in seeds])).cuda()
        self.cores = []
        for _ in range(ncontexts):
            if scaling == "strong":
                enc = fic_amd.ShardedEncoder(W, H, B, None, n_iso, planes, env.local_rank)
                c_, self.spans = enc.enc, enc.spans
            else:
                c_ = fic_amd.Encoder(W, H, B, None, n_iso, planes, env.local_rank)
                self.spans = [(0, c_.n_ranges)] * env.world
            c_.set_gray(self.dev_in)
            c_.set_option("time_sweep", 1)
            if chunks:
                c_.set_option("chunks", chunks)
            if sweep:
                c_.set_option("sweep", sweep)
            self.cores.append(c_)
        self.core = self.cores[0]
        self.begin, self.count = self.spans[env.rank]
        # The sweeps run on compute streams of their own; the codebook gather of step k runs on torch's current stream (where
        # the nccl backend orders its collectives) and overlaps with the sweep of step k+1.
        self.computes = [torch.cuda.Stream() for _ in self.cores]
        self.records = [c_.records_device() for c_ in self.cores]      # [planes, N_r, 6] int32 per context, written by its encodes
        self.nbuf = 2
        self.stage = [None] * self.nbuf      # copies of this rank's span where it is not contiguous in `records`
        self.staged = [None] * self.nbuf
        self.gathered = [None] * self.nbuf
        self.step_no = 0
        self.depth = 1
        self.Nr, self.Nd = self.core.n_ranges, self.core.n_domains
        self.ranges_per_step_rank = self.count * planes
        self.ranges_per_step_total = self.Nr * planes * (env.world if scaling == "weak" else 1)

    def make_image(self, s, w=None, h=None):
        fic_amd, np = self.env.fic_amd, self.env.np
        w, h = w or self.wl["W"], h or self.wl["H"]
        if self.dist_kind == "U":
            return fic_amd.synth.image_u(w, h, s)
        if self.dist_kind == "S":
            return fic_amd.synth.image_s(w, h, s)
        base = np.load(os.path.join(ROOT, "tests", "golden", "lena_grey_256.npy"))
        t = np.tile(base, ((h + 511) // 256 + 1, (w + 511) // 256 + 1))
        oy, ox = (s * 7) % 256, (s * 13) % 256
        return np.ascontiguousarray(t[oy:oy + h, ox:ox + w])

    def set_option(self, name, value):
        for c_ in self.cores:
            c_.set_option(name, value)

    def setup(self):
        """Not a warm-up step: the first encode of a context allocates its fragment / queue buffers and makes the runtime load
        the code object (8 ms against 2 ms per step).  Done once per context before the W warm-up steps so that `--warmup 0`
        still times steady-state steps; no gather, nothing timed."""
        for c_, st_ in zip(self.cores, self.computes):
            c_.encode(self.begin, self.count, st_)
        self.env.torch.cuda.synchronize()
        self.sweep_times()

    def step(self):
        torch, env = self.env.torch, self.env
        depth = self.depth
        k = self.step_no % self.nbuf
        ci = self.step_no % depth
        self.step_no += 1
        c_, compute = self.cores[ci], self.computes[ci]
        if env.world == 1:
            c_.encode(self.begin, self.count, compute)
            return
        with torch.cuda.stream(compute):
            if self.gathered[k] is not None:
                compute.wait_event(self.gathered[k])      # the gather that last read this buffer / context has finished
            c_.encode(self.begin, self.count, compute)
            src = self.records[ci][:, self.begin:self.begin + self.count]
            if depth == self.nbuf and src.is_contiguous():
                self.stage[k] = src                       # the context's own records: rewritten only by its next encode
            else:
                self.stage[k] = src.clone() if self.stage[k] is None or self.stage[k].data_ptr() == src.data_ptr() else self.stage[k].copy_(src)
            self.staged[k] = torch.cuda.Event()
            self.staged[k].record(compute)
        cur = torch.cuda.current_stream()
        cur.wait_event(self.staged[k])
        rec = self.stage[k] if env.backend == "nccl" else self.stage[k].cpu()
        env.fic_amd.gather_records(rec, self.spans, None, 0)
        self.gathered[k] = torch.cuda.Event()
        self.gathered[k].record(cur)

    def sweep_times(self, reset=True):
        ms = n = 0
        for c_ in self.cores:                             # contexts that did not run report (0, 0)
            m_, n_ = c_.sweep_time(reset=reset)
            ms, n = ms + m_, n + n_
        return ms, n

    def timed(self, nsteps):
        """EXACTLY nsteps steps between barrier + synchronize on both sides; returns (seconds on this rank, summed HIP-event
        duration of the sweep launches, their number)."""
        torch = self.env.torch
        torch.cuda.synchronize()
        self.sweep_times()
        self.env.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            self.step()
        torch.cuda.synchronize()
        self.env.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = self.sweep_times()
        return dt, ms, n

    def verify(self):
        """After (and outside) the timed region: this rank's span re-encoded into a second context by the VALU-only sweep
        (k_sweep_d4 / k_sweep_fast: another kernel family, exact integer covariances on the vector ALUs), and the 24-byte
        codebook records compared on the device with what the timed kernel left behind."""
        torch, fic_amd, wl = self.env.torch, self.env.fic_amd, self.wl
        torch.cuda.synchronize()
        mine = self.records[0][:, self.begin:self.begin + self.count].clone()
        chk = fic_amd.Encoder(wl["W"], wl["H"], wl["B"], None, wl["n_iso"], wl["planes"], self.env.local_rank)
        try:
            valu = 5 if (wl["B"] == 8 and wl["n_iso"] == 8) else 2
            chk.set_option("sweep", valu)
            chk.set_gray(self.dev_in)
            chk.encode(self.begin, self.count, self.computes[0])
            chk.sync()
            other = chk.records_device()[:, self.begin:self.begin + self.count]
            same = bool(torch.equal(mine, other))
            kind = chk.info()["sweep_kind"]
            if not same:                      # say where: the line only carries true / false
                bad = (mine != other).any(dim=-1).nonzero()
                p0, r0 = int(bad[0][0]), int(bad[0][1])
                print(f"bench.py rank {self.env.rank}: verify: {bad.shape[0]} of {mine.shape[0] * mine.shape[1]} records differ "
                      f"(span {self.begin}+{self.count}); first: plane {p0} range {self.begin + r0}: timed kernel "
                      f"{mine[p0, r0].tolist()} vs {kernel_name(kind, wl['B'], wl['n_iso'])[0]} {other[p0, r0].tolist()}", file=sys.stderr, flush=True)
        finally:
            chk.close()
        ok = bool(self.env.min_over_ranks(1.0 if same else 0.0) > 0.5)
        return {"ok": ok, "against": kernel_name(kind, wl["B"], wl["n_iso"])[0], "records_compared_per_rank": int(mine.shape[0] * mine.shape[1]),
                "how": "after the timed region: the same span re-encoded by the VALU-only sweep into a second context; "
                       "24-byte records (index, a bits, b bits, isometry, quantised s, o) compared on the device, all ranks"}

    def close(self):
        self.env.torch.cuda.synchronize()
        for c_ in self.cores:
            c_.close()
        self.cores = []
        self.dev_in = None
        self.records = self.stage = None


def roofline_blocks(args, run, info, avg_ms, sweep_n, clock_ghz):
    """`roofline` (+ the logical-HBM block) of the sweep launches of one Run."""
    env, wl = run.env, run.wl
    B, n_iso, planes = wl["B"], wl["n_iso"], wl["planes"]
    n, Nd = B * B, run.Nd
    kind = info["sweep_kind"]
    kname, operand = kernel_name(kind, B, n_iso, info["chunks"])
    kname = run.core.last_kernel()                         # the library knows which instantiation it launched (k_sweep_q / k_sweep_q16)
    pair_evals = float(run.ranges_per_step_rank) * Nd * n_iso
    # SURVEY 8(d)'s byte model: n+8 bytes per (range, domain) pair, one pool read per range block
    alg_bytes = float(run.ranges_per_step_rank) * Nd * (n + 8)
    rppr = env.fic_amd.capi.lib().fic_sweep_ranges_per_pool_read(kind, B, n_iso)
    hbm_logical = {"bound": "hbm (logical)", "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
                   "ranges_per_pool_read": rppr or None,
                   "note": "SURVEY 8(d): ranges x N_d x (n+8) bytes per launch.  NOT the bound of this kernel: a wave keeps its "
                           "range tile in registers and reads a pool block once for all of it (`ranges_per_pool_read` range blocks per "
                           "wave, from fic_sweep_ranges_per_pool_read; the waves of a workgroup share the load through L1/L2), so the "
                           "physical traffic is `traffic` and the sweep is compute-bound (roofline)"}
    traffic, traffic_note = None, "no PMC pass recorded for this kernel/workload on this source tree (tools/gpu_traffic.sh writes profiles/traffic.json)"
    try:   # HBM-side bytes per sweep launch from a rocprofv3 --pmc side pass of this same command (profiles/traffic.json)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        ent = tj.get(f"{args.workload}:planes={planes}:n_iso={n_iso}:kernel={kname}:gpus={env.world}")
        if ent and ent.get("csrc_hash") == csrc_hash():
            traffic, traffic_note = ent["bytes_per_launch"], ent["how"]
        elif ent:
            traffic_note = f"stale: profiles/traffic.json was measured on csrc {ent.get('csrc_hash')}, this tree is {csrc_hash()}"
    except (OSError, ValueError, KeyError):
        pass
    if operand is not None:
        # matrix-core sweeps: algorithmic work = 2n flop per pair evaluation (n MACs per pair)
        peak = MFMA_PEAK_TFLOPS[operand]
        ops = pair_evals * 2.0 * n
        # k_sweep_q's folded mode (8 isometries at B = 8 / 16) gets the 8 inner products of a (range, domain) pair from
        # 4 even + 4 odd rows of K = n/2: half the matrix instructions of the algorithmic count
        executed = ops * (0.5 if (kind == 6 and n_iso == 8 and B >= 8) else 1.0)
        roofline = {"bound": "mfma", "achieved": ops / (avg_ms * 1e-3) / 1e12, "peak": peak,
                    "unit": "TFLOP/s" if operand != "i8" else "TOP/s", "frac": ops / (avg_ms * 1e-3) / 1e12 / peak,
                    "executed_frac": executed / (avg_ms * 1e-3) / 1e12 / peak,
                    "traffic": traffic, "traffic_note": traffic_note, "kernel": kname, "operands": operand,
                    "avg_launch_ms": avg_ms, "launches": sweep_n, "algorithmic_flop_per_launch": ops,
                    "executed_flop_per_launch": executed, "matrix_pipe_frac": executed / (avg_ms * 1e-3) / 1e12 / peak,
                    "clock_ghz": clock_ghz, "peak_clock_ghz": PEAK_CLOCK_GHZ,
                    "matrix_pipe_frac_at_clock": executed / (avg_ms * 1e-3) / 1e12 / (peak * clock_ghz / PEAK_CLOCK_GHZ) if clock_ghz else None,
                    "note": "achieved / frac: ALGORITHMIC flop = range blocks x N_d x n_iso x 2n per launch (SURVEY 8d: n MACs per "
                            "pair evaluation) over the HIP-event duration of the launch on the sweep's stream, inside the timed "
                            "region.  executed_frac (= matrix_pipe_frac): what the matrix cores really issue -- the folded 8-isometry "
                            "form needs half the MACs (DESIGN.md 4.1), so frac can be 2x executed_frac and is NOT a pipe utilisation; "
                            "executed_frac is the utilisation against the 2.4 GHz peak; clock_ghz is what the chip held under this kernel "
                            "(shader-clock cycles / 100 MHz ticks of its own waves, a separate pass outside the timed region) and "
                            "matrix_pipe_frac_at_clock the matrix-pipe utilisation at that clock"}
    else:
        # VALU instructions per pair evaluation of the sweep kernel (static count from the ISA):
        # k_sweep_fast: n/4 v_dot4 per pair evaluation, plus per (range, domain): 1 iso -> 5; 8 iso -> 15, shared by 8
        # k_sweep_d4: (n + n/2) / 2 v_dot2c + 37 other VALU per (range, domain), shared by the 8 copies (ISA count at B = 8)
        vpe = ((n + n // 2) / 2 + 37.0) / 8.0 if kind == 5 else n / 4 + (15.0 / 8.0 if n_iso == 8 else 5.0)
        ach = pair_evals * vpe / 64.0 / (avg_ms * 1e-3)
        roofline = {"bound": "valu", "achieved": ach / 1e9, "peak": VALU_WAVE_INSTR_PEAK / 1e9, "unit": "G wave-instr/s",
                    "frac": ach / VALU_WAVE_INSTR_PEAK, "traffic": traffic, "traffic_note": traffic_note, "kernel": kname,
                    "avg_launch_ms": avg_ms, "launches": sweep_n, "valu_instr_per_pair_eval": vpe,
                    "note": "peak = 1 wave64 VALU instruction / 4 cycles / SIMD x 1024 SIMDs x 2.4 GHz, the measured issue rate "
                            "of v_dot4_u32_u8 / v_dot2c_i32_i16 (profiles/r01_valu_issue_rate_microbench.txt)"}
    return roofline, hbm_logical, pair_evals


def measure_clock(run, nsteps):
    """The clock the chip holds under k_sweep_q, from the kernel's own counters (shader-clock cycles over 100 MHz ticks of a
    sample of its waves).  Dense MFMA work is power-limited well below the 2.4 GHz the peak figures assume."""
    core = run.core
    core.set_option("sweep_stats", 1)
    for _ in range(nsteps):
        core.encode(run.begin, run.count, run.computes[0])
    st = core.sweep_stats()
    core.set_option("sweep_stats", 0)
    return st


def extra_block(env, args, name, scaling, steps):
    """A few timed steps of another BASELINE configuration through the same Run (after the headline, outside its timed
    region): ms per encode, matches/s, the kernel and its roofline fractions, and the post-run codebook check."""
    wl = dict(WORKLOADS[name])
    if args.extra_size and wl["W"] > args.extra_size:          # tests only
        wl["W"] = wl["H"] = args.extra_size
        wl["desc"] += f" [shrunk to {args.extra_size}x{args.extra_size} by --extra-size]"
    env.watchdog.stage = f"extra block {name} ({scaling})"
    run = Run(env, wl, scaling, 1, args.sweep, 0, "U")
    try:
        run.setup()
        run.step()
        dt, sweep_ms, sweep_n = run.timed(steps)
        dt = env.max_over_ranks(dt)
        per_rank = env.gather_floats(sweep_ms / max(sweep_n, 1))
        info = run.core.info()
        ver = None if args.no_verify else run.verify()
        out = None
        if env.rank == 0:
            avg_ms = sweep_ms / max(sweep_n, 1)
            roofline, _, pair_evals = roofline_blocks(args, run, info, avg_ms, sweep_n, None)
            out = {"workload": wl["desc"], "scaling": scaling, "n_gpus": env.world, "steps": steps,
                   "ms_per_encode": dt / steps * 1e3, "value": run.ranges_per_step_total * steps / dt, "unit": "range-block matches/s",
                   "kernel": roofline["kernel"], "avg_launch_ms": avg_ms, "per_rank_sweep_ms": per_rank, "pool_chunks": info["chunks"],
                   "pair_evals_per_s": pair_evals / (avg_ms * 1e-3) if avg_ms > 0 else None,
                   "roofline": {k: roofline.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "executed_frac")},
                   "verified": ver["ok"] if ver else None, "verified_against": ver["against"] if ver else None,
                   "N_r": run.Nr, "N_d": run.Nd, "ranges_per_rank": run.count}
        return out
    finally:
        run.close()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.inproc:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing GPU-related has been imported yet.
        sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    watchdog = Watchdog(rank, bench_timeout_s())
    if args.probe:
        sys.exit(probe(args, world, rank))

    import numpy as np
    import torch
    import fic_amd

    wl = dict(WORKLOADS[args.workload])
    if args.size or args.block:                         # exploratory overrides (not a BASELINE.json configuration)
        wl["W"] = wl["H"] = args.size or wl["W"]
        wl["B"] = args.block or wl["B"]
        wl["desc"] = f"{args.workload} with overrides: {wl['W']}x{wl['H']} synthetic grey U, B={wl['B']}, full search"
    if args.planes:
        wl["planes"] = args.planes
    if args.n_iso:
        wl["n_iso"] = args.n_iso
    scaling = args.scaling or wl["scaling"]
    if not args.inproc:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.inproc:
        sys.exit(bench_inproc(args, wl))
    # FIC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # records travel through host memory); the real multi-GPU run uses nccl == RCCL over xGMI.
    backend = os.environ.get("FIC_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    if local_rank >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                 f"(FIC_BENCH_BACKEND=gloo rehearses more ranks than GPUs)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from datetime import timedelta
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        watchdog.stage = f"init_process_group({backend})"
        # a rendezvous that does not complete in two minutes will not complete: fail instead of hanging the job
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=timedelta(seconds=120))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(seconds=120))
    env = Env(args, world, rank, local_rank, backend, dist, fic_amd, torch, np, watchdog)

    W, H, B, n_iso, planes = wl["W"], wl["H"], wl["B"], wl["n_iso"], wl["planes"]
    # `--pipeline 2`: two contexts on two streams take the steps alternately, so the pool build / range prep of step k+1 and
    # the ragged tail of sweep k overlap on the chip (every step still does all of its work inside the timed region).  The
    # headline run keeps depth 1: with two sweeps sharing the chip a launch's HIP-event duration is no longer the time the
    # kernel needs, and `roofline` must come from the same timed region as `value`.  N = 1 reports depth 2 beside it.
    depth = max(1, min(2, args.pipeline))
    want_pipelined_leg = world == 1 and not args.no_alt and depth == 1
    watchdog.stage = "headline: set-up"
    run = Run(env, wl, scaling, 2 if want_pipelined_leg else depth, args.sweep, args.chunks, args.dist)
    run.depth = depth
    run.setup()
    watchdog.stage = "headline: warm-up + timed steps"
    for _ in range(max(args.warmup, depth) if args.warmup else 0):
        run.step()
    dt, sweep_ms, sweep_n = run.timed(args.steps)
    dt = env.max_over_ranks(dt)
    per_rank_sweep_ms = env.gather_floats(sweep_ms / max(sweep_n, 1))
    rccl_ranks = None
    if dist:
        rccl_ranks = dist.get_world_size() if backend == "nccl" else 0
    info = run.core.info()
    watchdog.stage = "headline: codebook check"
    verified = None if args.no_verify else run.verify()

    total_ranges = run.ranges_per_step_total * args.steps
    value = total_ranges / dt
    out = None
    if rank == 0:
        kind = info["sweep_kind"]
        clock_ghz = None
        if kind == 6:
            clock_ghz = measure_clock(run, max(5, min(args.steps, 20)))["clock_ghz"]
        avg_ms = sweep_ms / max(sweep_n, 1)
        roofline, hbm_logical, pair_evals = roofline_blocks(args, run, info, avg_ms, sweep_n, clock_ghz)
        operand = roofline.get("operands")
        out = {
            "metric": "range-block matches/sec (8x8 R, 16x16 D, 8 iso)" if (B == 8 and n_iso == 8) else
                      f"range-block matches/sec ({B}x{B} R, {2 * B}x{2 * B} D, {n_iso} iso)",
            "value": value, "unit": "range-block matches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8",
            "dtype_detail": {"bf16": "centred u8 pixels as exact bf16, v_mfma_f32_32x32x16_bf16 -> exact f32 covariances, f32 prune test, f64/f32 Java epilogue",
                             "i8": "u8 pixels shifted to i8, v_mfma_i32_32x32x32_i8 -> exact i32 covariances, same epilogue",
                             "f16": "centred u8 range pixels (exact f16) x normalised domain pixels (f16), v_mfma_f32_32x32x16_f16 (1 isometry at B = 8 / 16: 16x16x32) -> "
                                    "|cov|/sqrt(var) within a proven bound = the prune test; surviving pairs: exact integer covariance "
                                    "(v_dot4_u32_u8) + f64/f32 Java epilogue",
                             None: "u8 pixels, v_dot4_u32_u8 / v_dot2c_i32_i16 -> exact i32 covariances, f32 prune test, f64/f32 Java epilogue"}[operand],
            "data": "synthetic",
            "config": {"workload": wl["desc"] if args.dist == "U" else wl["desc"].replace("synthetic grey U", f"grey {args.dist}"),
                       "dist": args.dist, "image": f"{W}x{H}", "B": B, "n_iso": n_iso, "wK": run.core.wK,
                       "planes_per_rank" if scaling == "weak" else "planes": planes,
                       "N_r": run.Nr, "N_d": run.Nd, "pool_chunks": info["chunks"], "sweep_kind": kind, "pipeline_depth": depth,
                       "setup": "one untimed encode per context before the warm-up steps (buffer allocation, code object load)",
                       "parallelism": f"range/plane shards x{world}"},
            "pair_evals_per_s": pair_evals * sweep_n / (sweep_ms * 1e-3) if sweep_ms > 0 else None,
            "roofline": roofline,
            "roofline_hbm_logical": hbm_logical,
            "verified": verified["ok"] if verified else None,
            "verified_detail": verified,
            "rccl_ranks": rccl_ranks,
            "per_rank_sweep_ms": per_rank_sweep_ms,
            "csrc_hash": csrc_hash(),
        }

    # ---- legs beside the headline (never `value`) ------------------------------------------------------------------
    if world == 1 and not args.no_alt and args.sustain > 0:
        watchdog.stage = "sustained"
        out["sustained"] = sustained(run, args.sustain, value)
    if world == 1 and not args.no_alt and out["roofline"].get("operands") is not None:
        # north_star's literal design (VALU only, no MFMA) on the same workload and buffers: reported beside, never as `value`
        watchdog.stage = "valu_only"
        run.set_option("sweep", 5 if (B == 8 and n_iso == 8) else 2)
        for _ in range(max(args.warmup, depth)):
            run.step()
        dtv, msv, nv = run.timed(args.steps)
        iv = run.core.info()
        run.set_option("sweep", args.sweep)
        out["valu_only"] = {"how": "fic_ctx_set_option(ctx, \"sweep\", 5 or 2) / FIC_SWEEP=5", "kernel": kernel_name(iv["sweep_kind"], B, n_iso)[0],
                            "value": total_ranges / dtv, "unit": "range-block matches/s", "ms_per_step": dtv / args.steps * 1e3,
                            "avg_launch_ms": msv / max(nv, 1), "default_speedup": dtv / dt,
                            "note": "bit-identical codebooks (tests/test_gpu_bench_geometry.py, test_gpu_fullsize.py, `verified`); this is the "
                                    "sweep north_star describes (no MFMA).  Its premise -- reduction/bandwidth-bound -- does not hold: "
                                    "profiles/r01z_cfg2_default_pmc_summary.txt shows 0.7 % of HBM peak and VALU busy 93.5 %, so the "
                                    "library default is the matrix-core sweep"}
    if want_pipelined_leg:
        watchdog.stage = "pipelined"
        run.depth = 2
        for _ in range(max(args.warmup, 2)):
            run.step()
        dtp, msp, npl = run.timed(args.steps)
        run.depth = depth
        out["pipelined"] = {"how": "bench.py --pipeline 2: two fic_ctx on two streams take the steps alternately", "depth": 2,
                            "value": total_ranges / dtp, "unit": "range-block matches/s", "ms_per_step": dtp / args.steps * 1e3,
                            "avg_launch_ms": msp / max(npl, 1),
                            "note": "same work per step; the prep kernels of step k+1 fill the ragged tail of sweep k, so the "
                                    "launches overlap and avg_launch_ms is no longer the time one sweep needs (hence not `value`)"}
    plain = args.workload == "cfg2" and not (args.size or args.block or args.planes or args.n_iso)
    if world == 1 and not args.no_alt and plain:
        watchdog.stage = "single_image"
        out["single_image"] = single_image(fic_amd, torch, run.make_image(run.seed), B, n_iso, local_rank, args.sweep)
        # config 2 names LenaGrey: the same encode on a natural image (the 256x256 fixture enlarged bilinearly), checked against
        # the VALU-only sweep -- smooth blocks prune differently from iid bytes
        watchdog.stage = "natural_image"
        lena = np.load(os.path.join(ROOT, "tests", "golden", "lena_grey_256.npy"))
        nat = fic_amd.synth.enlarge(lena, wl["W"], wl["H"])
        ni = single_image(fic_amd, torch, nat, B, n_iso, local_rank, args.sweep, reps=100)
        ni["workload"] = f"one {wl['W']}x{wl['H']} natural grey image (LenaGrey 256x256 enlarged bilinearly), B={B}, full search, {n_iso} iso"
        a_ = fic_amd.encode_gray(nat, B, None, n_iso, device=local_rank, sweep=args.sweep)
        b_ = fic_amd.encode_gray(nat, B, None, n_iso, device=local_rank, sweep=5 if (B == 8 and n_iso == 8) else 2)
        ni["verified"] = bool(all((a_[k].view(np.uint32) == b_[k].view(np.uint32)).all() if a_[k].dtype == np.float32 else (a_[k] == b_[k]).all()
                                  for k in ("idx_local", "iso", "qrows", "a", "b", "err")))
        out["single_image"]["natural_image"] = ni
        if n_iso != 1:          # the reference algorithm (no isometries) on the same image
            watchdog.stage = "single_image_one_isometry"
            oi = single_image(fic_amd, torch, run.make_image(run.seed), B, 1, local_rank, args.sweep, reps=100)
            oi["workload"] = oi["workload"].replace("(BASELINE config 2 literally)", "(config 2's image through the reference algorithm: 1 isometry)")
            out["single_image"]["one_isometry"] = oi
    img0 = run.dev_in[0].cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    run.close()

    # ---- the other BASELINE configurations and the strong-scaling point -------------------------------------------------
    if plain and not args.no_extra and not (world == 1 and args.no_alt):
        strong = None
        if world == 1:
            others = {}
            for name in OTHER_CONFIGS:
                others[name] = extra_block(env, args, name, WORKLOADS[name]["scaling"], args.extra_steps)
            out["other_configs"] = others
            strong = dict(others["cfg4"], note="N = 1: the same run as other_configs.cfg4")
        else:
            strong = extra_block(env, args, "cfg4", "strong", args.extra_steps)
        if rank == 0:
            strong["what"] = ("BASELINE config 4 (one 4096x4096 image, B=8, full search, 8 isometries): range blocks sharded over the "
                              "ranks, pool replicated, one gather of 24-byte records per step -- north_star's strong-scaling point "
                              "(>= 6x at 8 GPUs = this block's `value` at N=8 over its `value` at N=1)")
            out["strong_cfg4"] = strong

    if rank == 0:
        watchdog.stage = "cpu_baseline"
        if img0 is not None:
            out["cpu_baseline"] = cpu_baseline(wl, img0, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    watchdog.stage = "final barrier"
    env.barrier()
    if dist:
        dist.destroy_process_group()
    watchdog.cancel()
    if verified is not None and not verified["ok"]:
        sys.exit("bench.py: the codebook of the timed configuration differs from the VALU-only sweep's (verified = false)")


def sustained(run, seconds, headline_value):
    """The same step loop for >= `seconds` of wall time (not --steps): what the chip sustains once its clock has settled
    under the load (a 40 ms timed region on a power-limited kernel starts from a cool chip).  Then, without a pause, the
    loop goes on with the kernel's own clock counters switched on."""
    torch = run.env.torch
    batch = 100
    torch.cuda.synchronize()
    run.sweep_times()
    t0 = time.perf_counter()
    steps = 0
    while True:
        for _ in range(batch):
            run.step()
        steps += batch
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    ms, n = run.sweep_times()
    clock = None
    kind = run.core.info()["sweep_kind"]
    if kind == 6:
        t1 = time.perf_counter()
        run.core.set_option("sweep_stats", 1)
        nclk = 0
        while time.perf_counter() - t1 < max(0.25 * seconds, 0.2):
            for _ in range(batch):
                run.core.encode(run.begin, run.count, run.computes[0])
            nclk += batch
            torch.cuda.synchronize()
        clock = run.core.sweep_stats()["clock_ghz"]
        run.core.set_option("sweep_stats", 0)
    v = run.ranges_per_step_total * steps / dt
    return {"seconds": dt, "steps": steps, "value": v, "unit": "range-block matches/s", "ms_per_step": dt / steps * 1e3,
            "avg_launch_ms": ms / max(n, 1), "clock_ghz": clock, "vs_timed_region": v / headline_value,
            "note": "same step() as the headline, run back to back for >= --sustain seconds of wall time (host sync every 100 "
                    "steps); clock_ghz: the in-kernel clock of k_sweep_q sampled while the loop simply continues"}


def single_image(fic_amd, torch, img, B, n_iso, device, sweep, reps=200):
    """The literal BASELINE config 2: ONE 512x512 image, resident in HBM.  `ms` = back-to-back encodes on one stream
    (what a caller streaming images sees); `latency_ms` = one encode + stream sync, host clock."""
    H, W = img.shape
    enc = fic_amd.Encoder(W, H, B, None, n_iso, 1, device)
    d = torch.from_numpy(img[None]).cuda()
    enc.set_gray(d)
    if sweep:
        enc.set_option("sweep", sweep)
    s = torch.cuda.Stream()
    for _ in range(5):
        enc.encode(0, -1, s)
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        enc.encode(0, -1, s)
    e1.record(s)
    s.synchronize()
    ms = e0.elapsed_time(e1) / reps
    lat = []
    for _ in range(50):
        t0 = time.perf_counter()
        enc.encode(0, -1, s)
        s.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    enc.set_option("time_sweep", 1)
    for _ in range(20):
        enc.encode(0, -1, s)
    sw_ms, sw_n = enc.sweep_time(reset=True)
    info = enc.info()
    kname_single = enc.last_kernel()
    nr = enc.n_ranges
    enc.close()
    return {"workload": f"one {W}x{H} synthetic grey U image, B={B}, full search, {n_iso} iso (BASELINE config 2 literally)",
            "ms": ms, "matches_per_s": nr / (ms * 1e-3), "latency_ms": lat[len(lat) // 2],
            "sweep_ms": sw_ms / max(sw_n, 1), "kernel": kname_single, "pool_chunks": info["chunks"],
            "note": "ms: HIP events around 200 back-to-back encodes (pool build + range prep + sweep + finalise each); "
                    "latency_ms: median host time of encode + sync"}


def bench_inproc(args, wl):
    """--inproc: the in-library multi-device entry (one process, one host thread, args.gpus devices, RCCL gather
    inside the library) -- what the JNI host calls.  Host buffers in and out, so this figure includes PCIe."""
    import numpy as np
    import fic_amd
    from fic_amd import capi
    W, H, B, n_iso = wl["W"], wl["H"], wl["B"], wl["n_iso"]
    img = fic_amd.synth.image_u(W, H, fic_amd.synth.SEEDS[wl["seed"]])
    for _ in range(max(args.warmup, 1)):
        capi.encode_gray_multi(img, B, None, n_iso, args.gpus)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = capi.encode_gray_multi(img, B, None, n_iso, args.gpus)
    dt = time.perf_counter() - t0
    nr = r["idx_local"].size
    print(json.dumps({"metric": f"range-block matches/sec ({B}x{B} R, {2 * B}x{2 * B} D, {n_iso} iso), in-library multi-device entry",
                      "value": nr * args.steps / dt, "unit": "range-block matches/s", "n_gpus": args.gpus, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                      "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                      "config": {"workload": wl["desc"] + " -- fic_encode_gray_u8_multi, host buffers in/out (PCIe inclusive)"}}))
    return 0


if __name__ == "__main__":
    main()
