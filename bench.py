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
children, spawned before this process touches torch or the GPU; it waits for them and exits
with their worst code); under `python -m torch.distributed.run ... bench.py --gpus N` the
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment is used as given.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the compute
bound of the sweep kernel that ran: matrix-core FLOP/s against the dense MFMA peak, or VALU
issue for the VALU-only sweeps), `roofline_hbm_logical` (SURVEY 8d's byte model, which the
sweep exceeds by the register-tile reuse factor), `valu_only` (the same workload through the
VALU-only sweep north_star describes), `single_image` and `cpu_baseline` (the C oracle, 1 core,
bounded sample).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
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
        # folded; MULTI: the launch has more than one pool chunk
        return f"k_sweep_q<{B * B // 16}, {0 if n_iso == 1 else (1 if B == 4 else 2)}, {'true' if chunks > 1 else 'false'}>", "f16"
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


def launch_ranks(n, cmd, env=None, poll_s=0.2):
    """Starts `cmd` n times as fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set (one process per GPU; rendezvous on 127.0.0.1), waits for all of them and returns the worst exit code.  If a
    rank fails, the others are stopped (their exact PIDs) instead of being left at a barrier.  The calling process
    must not have touched the GPU: nothing here imports torch or loads libfic_hip.so."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=e))
    worst = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                for o in sorted(live):            # a failed rank leaves the others waiting at a collective
                    procs[o].terminate()
        if live:
            time.sleep(poll_s)
            if worst:
                deadline = time.time() + 10.0
                while live and time.time() < deadline:
                    live = {r for r in live if procs[r].poll() is None}
                    time.sleep(poll_s)
                for o in sorted(live):
                    procs[o].kill()
                    procs[o].wait()
                live = set()
    return worst


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
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
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
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timings (valu_only, single_image)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--inproc", action="store_true",
                    help="time the in-library multi-device entry fic_encode_gray_u8_multi (one process, --gpus devices) instead")
    ap.add_argument("--probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-fail-rank", type=int, default=None, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.inproc:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing GPU-related has been imported yet.
        sys.exit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if dist:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()

    W, H, B, n_iso, planes = wl["W"], wl["H"], wl["B"], wl["n_iso"], wl["planes"]
    seed = fic_amd.synth.SEEDS[wl["seed"]]
    # synthetic input, generated once and uploaded: resident in HBM before any timing
    def make_image(s, w=W, h=H):
        if args.dist == "U":
            return fic_amd.synth.image_u(w, h, s)
        if args.dist == "S":
            return fic_amd.synth.image_s(w, h, s)
        base = np.load(os.path.join(ROOT, "tests", "golden", "lena_grey_256.npy"))
        t = np.tile(base, ((h + 511) // 256 + 1, (w + 511) // 256 + 1))
        oy, ox = (s * 7) % 256, (s * 13) % 256
        return np.ascontiguousarray(t[oy:oy + h, ox:ox + w])

    if scaling == "weak":
        imgs = np.stack([make_image(seed + 3 * (rank * planes + p)) for p in range(planes)])
    else:
        imgs = np.stack([make_image(seed + 3 * p) for p in range(planes)])
    dev_in = torch.from_numpy(imgs).cuda()

    # `--pipeline 2`: two contexts on two streams take the steps alternately, so the pool build / range prep of step k+1 and
    # the ragged tail of sweep k overlap on the chip (every step still does all of its work inside the timed region).  The
    # headline run keeps depth 1: with two sweeps sharing the chip a launch's HIP-event duration is no longer the time the
    # kernel needs, and `roofline` must come from the same timed region as `value`.  N = 1 reports depth 2 beside it.
    depth = max(1, min(2, args.pipeline))
    want_pipelined_leg = world == 1 and not args.no_alt and depth == 1
    cores = []
    for _ in range(2 if want_pipelined_leg else depth):
        if scaling == "strong":
            enc = fic_amd.ShardedEncoder(W, H, B, None, n_iso, planes, local_rank)
            c_, spans = enc.enc, enc.spans
        else:
            c_ = fic_amd.Encoder(W, H, B, None, n_iso, planes, local_rank)
            spans = [(0, c_.n_ranges)] * world
        c_.set_gray(dev_in)
        c_.set_option("time_sweep", 1)
        if args.chunks:
            c_.set_option("chunks", args.chunks)
        if args.sweep:
            c_.set_option("sweep", args.sweep)
        cores.append(c_)
    core = cores[0]
    begin, count = spans[rank]
    # The sweeps run on compute streams of their own; the codebook gather of step k runs on torch's current stream (where
    # the nccl backend orders its collectives) and overlaps with the sweep of step k+1.
    computes = [torch.cuda.Stream() for _ in cores]
    records = [c_.records_device() for c_ in cores]       # [planes, N_r, 6] int32 per context, written by its encodes
    nbuf = 2
    stage = [None] * nbuf                                 # copies of this rank's span where it is not contiguous in `records`
    staged = [None] * nbuf
    gathered = [None] * nbuf
    step_no = [0]
    active_depth = [depth]

    def step():
        depth = active_depth[0]
        k = step_no[0] % nbuf
        ci = step_no[0] % depth
        step_no[0] += 1
        c_, compute = cores[ci], computes[ci]
        if world == 1:
            c_.encode(begin, count, compute)
            return
        with torch.cuda.stream(compute):
            if gathered[k] is not None:
                compute.wait_event(gathered[k])           # the gather that last read this buffer / context has finished
            c_.encode(begin, count, compute)
            src = records[ci][:, begin:begin + count]
            if depth == nbuf and src.is_contiguous():
                stage[k] = src                            # the context's own records: rewritten only by its next encode
            else:
                stage[k] = src.clone() if stage[k] is None or stage[k].data_ptr() == src.data_ptr() else stage[k].copy_(src)
            staged[k] = torch.cuda.Event()
            staged[k].record(compute)
        cur = torch.cuda.current_stream()
        cur.wait_event(staged[k])
        rec = stage[k] if backend == "nccl" else stage[k].cpu()
        fic_amd.gather_records(rec, spans, None, 0)
        gathered[k] = torch.cuda.Event()
        gathered[k].record(cur)

    def sweep_times(reset=True):
        ms = n = 0
        for c_ in cores:                                  # contexts that did not run report (0, 0)
            m_, n_ = c_.sweep_time(reset=reset)
            ms, n = ms + m_, n + n_
        return ms, n

    def timed(nsteps):
        torch.cuda.synchronize()
        sweep_times()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = sweep_times()
        return dt, ms, n

    # Set-up, not a warm-up step: the first encode of a context allocates its fragment / queue buffers and makes the runtime load
    # the code object (8 ms against 2 ms per step).  Done once per context before the W warm-up steps so that `--warmup 0`
    # still times steady-state steps; no gather, nothing timed.
    for c_, st_ in zip(cores, computes):
        c_.encode(begin, count, st_)
    torch.cuda.synchronize()
    sweep_times()
    for _ in range(max(args.warmup, depth) if args.warmup else 0):
        step()
    dt, sweep_ms, sweep_n = timed(args.steps)
    per_rank_sweep_ms = [sweep_ms / max(sweep_n, 1)]
    rccl_ranks = None
    if dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([sweep_ms / max(sweep_n, 1)], dtype=torch.float64, device=dev)
        allms = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allms, mine)
        per_rank_sweep_ms = [float(x.item()) for x in allms]
        rccl_ranks = dist.get_world_size() if backend == "nccl" else 0

    Nr, Nd, n = core.n_ranges, core.n_domains, B * B
    ranges_per_step_rank = count * planes
    total_ranges = Nr * planes * (world if scaling == "weak" else 1) * args.steps
    value = total_ranges / dt
    info = core.info()

    if rank == 0:
        kind = info["sweep_kind"]
        kname, operand = kernel_name(kind, B, n_iso, info["chunks"])
        # The clock the chip holds under this sweep (outside the timed region): k_sweep_q's own counters, shader-clock cycles
        # over 100 MHz ticks of a sample of its waves.  Dense MFMA work is power-limited well below the 2.4 GHz the peak
        # figures assume, so the roofline block reports the fraction against both.
        clock_ghz = None
        if kind == 6:
            core.set_option("sweep_stats", 1)
            for _ in range(max(5, min(args.steps, 20))):
                core.encode(begin, count, computes[0])
            clock_ghz = core.sweep_stats()["clock_ghz"]
            core.set_option("sweep_stats", 0)
        avg_ms = sweep_ms / max(sweep_n, 1)
        pair_evals = float(ranges_per_step_rank) * Nd * n_iso
        # SURVEY 8(d)'s byte model: n+8 bytes per (range, domain) pair, one pool read per range block
        alg_bytes = float(ranges_per_step_rank) * Nd * (n + 8)
        hbm_logical = {"bound": "hbm (logical)", "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": alg_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
                       "ranges_per_pool_read": 64 if operand is None else 16,
                       "note": "SURVEY 8(d): ranges x N_d x (n+8) bytes per launch.  NOT the bound of this kernel: a wave keeps its "
                               "range tile in registers and reads a pool block once for all of it (`ranges_per_pool_read` per wave, "
                               "four waves of a workgroup share the load through L1), so the physical traffic is `traffic` and the "
                               "sweep is compute-bound (roofline)"}
        traffic, traffic_note = None, "no PMC pass recorded for this kernel/workload on this source tree (tools/gpu_traffic.sh writes profiles/traffic.json)"
        try:   # HBM-side bytes per sweep launch from a rocprofv3 --pmc side pass of this same command (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"{args.workload}:planes={planes}:n_iso={n_iso}:kernel={kname}:gpus={world}")
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
                        "traffic": traffic, "traffic_note": traffic_note, "kernel": kname, "operands": operand,
                        "avg_launch_ms": avg_ms, "launches": sweep_n, "algorithmic_flop_per_launch": ops,
                        "executed_flop_per_launch": executed, "matrix_pipe_frac": executed / (avg_ms * 1e-3) / 1e12 / peak,
                        "clock_ghz": clock_ghz, "peak_clock_ghz": PEAK_CLOCK_GHZ,
                        "matrix_pipe_frac_at_clock": executed / (avg_ms * 1e-3) / 1e12 / (peak * clock_ghz / PEAK_CLOCK_GHZ) if clock_ghz else None,
                        "note": "achieved / frac: ALGORITHMIC flop = range blocks x N_d x n_iso x 2n per launch (SURVEY 8d: n MACs per "
                                "pair evaluation) over the HIP-event duration of the launch on the sweep's stream, inside the timed "
                                "region.  executed_flop / matrix_pipe_frac: what the matrix cores really issue -- the folded 8-isometry "
                                "form needs half the MACs (DESIGN.md 4.5), so frac can exceed matrix_pipe_frac by 2x; matrix_pipe_frac "
                                "is the utilisation against the 2.4 GHz peak; clock_ghz is what the chip held under this kernel "
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
        out = {
            "metric": "range-block matches/sec (8x8 R, 16x16 D, 8 iso)" if (B == 8 and n_iso == 8) else
                      f"range-block matches/sec ({B}x{B} R, {2 * B}x{2 * B} D, {n_iso} iso)",
            "value": value, "unit": "range-block matches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8",
            "dtype_detail": {"bf16": "centred u8 pixels as exact bf16, v_mfma_f32_32x32x16_bf16 -> exact f32 covariances, f32 prune test, f64/f32 Java epilogue",
                             "i8": "u8 pixels shifted to i8, v_mfma_i32_32x32x32_i8 -> exact i32 covariances, same epilogue",
                             "f16": "centred u8 range pixels (exact f16) x normalised domain pixels (f16), v_mfma_f32_32x32x16_f16 -> "
                                    "|cov|/sqrt(var) within a proven bound = the prune test; surviving pairs: exact integer covariance "
                                    "(v_dot4_u32_u8) + f64/f32 Java epilogue",
                             None: "u8 pixels, v_dot4_u32_u8 / v_dot2c_i32_i16 -> exact i32 covariances, f32 prune test, f64/f32 Java epilogue"}[operand],
            "data": "synthetic",
            "config": {"workload": wl["desc"] if args.dist == "U" else wl["desc"].replace("synthetic grey U", f"grey {args.dist}"),
                       "dist": args.dist, "image": f"{W}x{H}", "B": B, "n_iso": n_iso, "wK": core.wK,
                       "planes_per_rank" if scaling == "weak" else "planes": planes,
                       "N_r": Nr, "N_d": Nd, "pool_chunks": info["chunks"], "sweep_kind": kind, "pipeline_depth": depth,
                       "setup": "one untimed encode per context before the warm-up steps (buffer allocation, code object load)",
                       "parallelism": f"range/plane shards x{world}"},
            "pair_evals_per_s": pair_evals * sweep_n / (sweep_ms * 1e-3) if sweep_ms > 0 else None,
            "roofline": roofline,
            "roofline_hbm_logical": hbm_logical,
            "rccl_ranks": rccl_ranks,
            "per_rank_sweep_ms": per_rank_sweep_ms,
            "csrc_hash": csrc_hash(),
        }
        if world == 1 and not args.no_alt and operand is not None:
            # north_star's literal design (VALU only, no MFMA) on the same workload and buffers: reported beside, never as `value`
            for c_ in cores:
                c_.set_option("sweep", 5 if (B == 8 and n_iso == 8) else 2)
            for _ in range(max(args.warmup, depth)):
                step()
            dtv, msv, nv = timed(args.steps)
            iv = core.info()
            for c_ in cores:
                c_.set_option("sweep", args.sweep)
            out["valu_only"] = {"how": "fic_ctx_set_option(ctx, \"sweep\", 5 or 2) / FIC_SWEEP=5", "kernel": kernel_name(iv["sweep_kind"], B, n_iso)[0],
                                "value": total_ranges / dtv, "unit": "range-block matches/s", "ms_per_step": dtv / args.steps * 1e3,
                                "avg_launch_ms": msv / max(nv, 1), "default_speedup": dtv / dt,
                                "note": "bit-identical codebooks (tests/test_gpu_mfma.py, test_gpu_fullsize.py); this is the sweep "
                                        "north_star describes (no MFMA).  Its premise -- reduction/bandwidth-bound -- does not hold: "
                                        "profiles/r01z_cfg2_default_pmc_summary.txt shows 0.7 % of HBM peak and VALU busy 93.5 %, so the "
                                        "library default is the matrix-core sweep"}
        if want_pipelined_leg:
            active_depth[0] = 2
            for _ in range(max(args.warmup, 2)):
                step()
            dtp, msp, npl = timed(args.steps)
            active_depth[0] = depth
            out["pipelined"] = {"how": "bench.py --pipeline 2: two fic_ctx on two streams take the steps alternately", "depth": 2,
                                "value": total_ranges / dtp, "unit": "range-block matches/s", "ms_per_step": dtp / args.steps * 1e3,
                                "avg_launch_ms": msp / max(npl, 1),
                                "note": "same work per step; the prep kernels of step k+1 fill the ragged tail of sweep k, so the "
                                        "launches overlap and avg_launch_ms is no longer the time one sweep needs (hence not `value`)"}
        if world == 1 and not args.no_alt and args.workload == "cfg2" and not (args.size or args.block or args.planes or args.n_iso):
            out["single_image"] = single_image(fic_amd, torch, make_image(seed), B, n_iso, local_rank, args.sweep)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, imgs[0], args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    barrier()
    if dist:
        dist.destroy_process_group()


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
    nr = enc.n_ranges
    enc.close()
    return {"workload": f"one {W}x{H} synthetic grey U image, B={B}, full search, {n_iso} iso (BASELINE config 2 literally)",
            "ms": ms, "matches_per_s": nr / (ms * 1e-3), "latency_ms": lat[len(lat) // 2],
            "sweep_ms": sw_ms / max(sw_n, 1), "kernel": kernel_name(info["sweep_kind"], B, n_iso, info["chunks"])[0], "pool_chunks": info["chunks"],
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
