#!/usr/bin/env python3
"""bench.py -- range-block matches/s of the MI355X grey encode hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]

One "step" = one pass of the hot path (pool build + range prep + sweep + finalise + codebook
gather) over one batch of synthetic grey input that is already resident in HBM.  Default
workload = BASELINE.json configs[1], the configuration the metric is quoted on:
512x512 grey, 8x8 range / 16x16 domain (B=8), full search, 8 isometries -- as a batch of 64
images per GPU so a step is milliseconds, not launch latency.  At N>1 every rank encodes its
own batch (weak scaling: units = images, no data-path collective) and the codebooks are
gathered to rank 0 with one RCCL gather per step (the path's only exchange).
`--workload cfg4 --scaling strong` shards ONE 4096x4096 image's range blocks across the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel = the pool sweep; algorithmic bytes = ranges x N_d x (n+8) per launch) and
`cpu_baseline` (the C oracle, 1 core, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

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
# Measured on MI355X (profiles/r01_valu_issue_rate_microbench.txt): v_dot4_u32_u8 -- like v_mad_i32_i24,
# v_cvt_*, v_cmp_* and any VALU op with an SGPR source -- issues one wave64 instruction per 4 cycles
# per SIMD (only plain v_fma_f32 / v_add_u32 reach 2 cycles).  Peak for this kernel's instruction mix:
VALU_WAVE_INSTR_PEAK = 256 * 4 * 2.4e9 / 4.0   # wave-instructions/s, chip-wide, at the 2.4 GHz max clock


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


def main():
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
    ap.add_argument("--sweep", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="0/2 = VALU sweep k_sweep_fast (default, north_star's design); 3 = opt-in matrix-core sweep "
                         "(bf16 operands; i8 at B = 16 with 1 isometry); 4 = matrix-core sweep with i8 operands at every B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timing of the opt-in matrix-core sweep")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    # FIC_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share devices,
    # records travel through host memory); the real multi-GPU run uses nccl == RCCL over xGMI.
    backend = os.environ.get("FIC_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W, H, B, n_iso, planes = wl["W"], wl["H"], wl["B"], wl["n_iso"], wl["planes"]
    seed = fic_amd.synth.SEEDS[wl["seed"]]
    # synthetic input, generated once and uploaded: resident in HBM before any timing
    def make_image(s):
        if args.dist == "U":
            return fic_amd.synth.image_u(W, H, s)
        if args.dist == "S":
            return fic_amd.synth.image_s(W, H, s)
        base = np.load(os.path.join(ROOT, "tests", "golden", "lena_grey_256.npy"))
        t = np.tile(base, ((H + 511) // 256 + 1, (W + 511) // 256 + 1))
        oy, ox = (s * 7) % 256, (s * 13) % 256
        return np.ascontiguousarray(t[oy:oy + H, ox:ox + W])

    if scaling == "weak":
        imgs = np.stack([make_image(seed + 3 * (rank * planes + p)) for p in range(planes)])
    else:
        imgs = np.stack([make_image(seed + 3 * p) for p in range(planes)])
    dev_in = torch.from_numpy(imgs).cuda()

    if scaling == "strong":
        enc = fic_amd.ShardedEncoder(W, H, B, None, n_iso, planes, local_rank)
        core = enc.enc
        begin, count = enc.spans[rank]
    else:
        core = fic_amd.Encoder(W, H, B, None, n_iso, planes, local_rank)
        begin, count = 0, core.n_ranges
        spans = [(0, core.n_ranges)] * world
    core.set_gray(dev_in)
    core.set_option("time_sweep", 1)
    if args.chunks:
        core.set_option("chunks", args.chunks)
    if args.sweep:
        core.set_option("sweep", args.sweep)
    stream = torch.cuda.current_stream()
    res_dev = core.results_device()

    def step():
        core.encode(begin, count, stream)
        if world > 1:
            rec = fic_amd.pack_records(res_dev, begin, count)
            if backend != "nccl":
                rec = rec.cpu()
            if scaling == "strong":
                fic_amd.gather_records(rec, enc.spans, None, 0)
            else:
                fic_amd.gather_records(rec, spans, None, 0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    core.sweep_time(reset=True)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sweep_ms, sweep_n = core.sweep_time(reset=True)
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    Nr, Nd, n = core.n_ranges, core.n_domains, B * B
    ranges_per_step_rank = count * planes
    if scaling == "weak":
        total_ranges = Nr * planes * world * args.steps
    else:
        total_ranges = Nr * planes * args.steps
    value = total_ranges / dt
    info = core.info()

    if rank == 0:
        avg_ms = sweep_ms / max(sweep_n, 1)
        alg_bytes = float(ranges_per_step_rank) * Nd * (n + 8)      # n+8 bytes per (range,domain) pair
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        pair_evals = float(ranges_per_step_rank) * Nd * n_iso
        # VALU instructions per pair evaluation of the sweep kernel (static count from the ISA):
        # n/4 v_dot4 per pair evaluation, plus per (range, domain): 1 iso -> 2 (base) + mul + cvt + cmp = 5;
        # 8 iso -> 2 (base) + mul + cvt + 8 (max3/min3 reduction of the 8 copies) + sub + 2 cmp = 15, shared by 8
        valu_per_eval = n / 4 + (15.0 / 8.0 if n_iso == 8 else 5.0)
        kernel_name = "k_sweep_fast"
        if info["sweep_kind"] == 5:
            # k_sweep_d4: (n + n/2) / 2 v_dot2c + 37 other VALU per (range, domain), shared by the 8 copies (ISA count at B = 8)
            valu_per_eval = ((n + n // 2) / 2 + 37.0) / 8.0
            kernel_name = "k_sweep_d4"
        valu_frac = pair_evals * valu_per_eval / 64.0 / (avg_ms * 1e-3) / VALU_WAVE_INSTR_PEAK
        traffic, traffic_note = None, None
        try:   # HBM-side bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/traffic.json)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"{args.workload}:planes={planes}:n_iso={n_iso}:sweep={info['sweep_kind']}:gpus={world}")
            if ent:
                traffic, traffic_note = ent["bytes_per_launch"], ent["how"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "range-block matches/sec (8x8 R, 16x16 D, 8 iso)" if (B == 8 and n_iso == 8) else
                      f"range-block matches/sec ({B}x{B} R, {2 * B}x{2 * B} D, {n_iso} iso)",
            "value": value, "unit": "range-block matches/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "u8",
            "dtype_detail": ("u8 pixels -> i16 group-Fourier slots of the 8 isometries, v_dot2c_i32_i16 -> exact i32 covariances, "
                             "f32 prune test, f64/f32 Java epilogue") if info["sweep_kind"] == 5 else
                            "u8 pixels, v_dot4_u32_u8 -> exact i32 covariances, f32 prune test, f64/f32 Java epilogue",
            "data": "synthetic",
            "config": {"workload": wl["desc"] if args.dist == "U" else wl["desc"].replace("synthetic grey U", f"grey {args.dist}"),
                       "dist": args.dist, "image": f"{W}x{H}", "B": B, "n_iso": n_iso, "wK": core.wK,
                       "planes_per_rank" if scaling == "weak" else "planes": planes,
                       "N_r": Nr, "N_d": Nd, "pool_chunks": info["chunks"], "parallelism": f"range/plane shards x{world}"},
            "pair_evals_per_s": pair_evals * sweep_n / (sweep_ms * 1e-3) if sweep_ms > 0 else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": kernel_name, "avg_launch_ms": avg_ms, "launches": sweep_n,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = ranges x N_d x (n+8); each wave keeps 64 range blocks in VGPRs "
                                 "and reads a pool block once for all of them, so frac > 1 means the sweep is past the "
                                 "HBM roofline and bounded by VALU issue instead (see valu)"},
            "valu": {"bound": "valu-issue (the true bound of this kernel)", "frac": valu_frac,
                     "valu_instr_per_pair_eval": valu_per_eval, "peak_wave_instr_per_s": VALU_WAVE_INSTR_PEAK,
                     "note": "peak = 1 wave64 VALU instruction / 4 cycles / SIMD x 1024 SIMDs x 2.4 GHz, the measured "
                             "issue rate of v_dot4_u32_u8 and v_dot2c_i32_i16 (profiles/r01_valu_issue_rate_microbench.txt); "
                             "about half of k_sweep_d4's 37 non-dot instructions are 2-cycle adds, so its frac is an upper estimate; PMC: SQ_ACTIVE_INST_VALU = 93.5% of kernel cycles (profiles/r01z_cfg2_default_pmc_summary.txt)"
                             if info["sweep_kind"] == 5 else
                             "peak = 1 wave64 VALU instruction / 4 cycles / SIMD x 1024 SIMDs x 2.4 GHz, the measured "
                             "issue rate of v_dot4_u32_u8 (profiles/r01_valu_issue_rate_microbench.txt); PMC: "
                             "SQ_ACTIVE_INST_VALU = 96% of kernel cycles (profiles/r01a_cfg2_pmc_summary.txt)"},
        }
        if info["sweep_kind"] in (3, 4):
            # opt-in matrix-core sweeps.  "sweep" = 3 at B = 4/8 (and B = 16 with 8 isometries): centred pixels as exact bf16 operands of
            # v_mfma_f32_32x32x16_bf16 (dense bf16 peak 2.5 PFLOP/s); otherwise u8 shifted to i8 on
            # v_mfma_i32_32x32x32_i8 (dense i8 peak 5.0 PetaOP/s) -- MI355X_MICROARCH.md "Matrix cores".
            bf16 = info["sweep_kind"] == 3 and (B <= 8 or n_iso == 8)
            kname = ("k_sweep_bf16" if bf16 else "k_sweep_mfma") + ("" if n_iso == 8 else ("_1" if bf16 else "1"))
            peak = 2500.0 if bf16 else 5000.0
            ops = pair_evals * 2.0 * n
            out["dtype_detail"] = ("centred u8 pixels as exact bf16, v_mfma_f32_32x32x16_bf16 -> exact f32 covariances, same epilogue"
                                   if bf16 else
                                   "u8 pixels shifted to i8, v_mfma_i32_32x32x32_i8 -> exact i32 covariances, same epilogue")
            out["roofline_hbm_logical"] = dict(out["roofline"], kernel=kname)
            out["roofline"] = {"bound": "mfma", "achieved": ops / (avg_ms * 1e-3) / 1e12, "peak": peak,
                               "unit": "TFLOP/s" if bf16 else "TOP/s", "frac": ops / (avg_ms * 1e-3) / 1e12 / peak,
                               "traffic": None, "kernel": kname, "avg_launch_ms": avg_ms, "launches": sweep_n,
                               "note": "opt-in (--sweep 3/4): north_star rules MFMA out for this path; default is the VALU sweep"}
            # bf16 kernels: 28 (8 iso) / 27 (1 iso) VALU wave-instructions per 32x32 tile of pair evaluations (ISA count)
            vpe = (28.0 * 64 / 1024 if n_iso == 8 else 27.0 * 64 / 1024) if bf16 else (3.5 if n_iso == 8 else 6.0)
            out["valu"] = {"bound": "valu-issue (epilogue)", "valu_instr_per_pair_eval": vpe,
                           "frac": pair_evals * vpe / 64.0 / (avg_ms * 1e-3) / VALU_WAVE_INSTR_PEAK,
                           "peak_wave_instr_per_s": VALU_WAVE_INSTR_PEAK}
        if info["sweep_kind"] in (2, 5) and world == 1 and not args.no_alt:
            # Same workload, same buffers, through the opt-in matrix-core sweep: reported beside, never as `value`.
            core.set_option("sweep", 3)
            for _ in range(args.warmup):
                step()
            torch.cuda.synchronize()
            core.sweep_time(reset=True)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
            ms3, n3 = core.sweep_time(reset=True)
            core.set_option("sweep", args.sweep)
            out["opt_in_matrix_core"] = {
                "how": "bench.py --sweep 3 / fic_ctx_set_option(ctx, \"sweep\", 3)",
                "kernel": (("k_sweep_bf16" if n_iso == 8 else "k_sweep_bf16_1") if (B <= 8 or n_iso == 8) else "k_sweep_mfma1"),
                "value": total_ranges / dt3, "unit": "range-block matches/s", "ms_per_step": dt3 / args.steps * 1e3,
                "avg_launch_ms": ms3 / max(n3, 1), "speedup_vs_default": dt / dt3,
                "mfma_roofline_frac": (pair_evals * 2.0 * n) / (ms3 / max(n3, 1) * 1e-3) / 1e12 /
                                      (2500.0 if (B <= 8 or n_iso == 8) else 5000.0),
                "note": "bit-identical codebooks (tests/test_gpu_mfma.py); inner products on v_mfma_f32_32x32x16_bf16 "
                        "(centred pixels are exact bf16) or, at B = 16 with 1 isometry, v_mfma_i32_32x32x32_i8. "
                        "Not the default because north_star asks for a VALU-only sweep (DESIGN.md section 6)."}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, imgs[0], args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
