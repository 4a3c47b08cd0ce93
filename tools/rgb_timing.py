#!/usr/bin/env python3
"""Timing of the joint-RGB encoder (fic_encode_rgb_argb = FractalCompression.encodeRGB, FC:171-219) through the one-shot
C entry point: host ARGB in, host codebook out.  Synthetic colour images: three U planes."""
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()          # torch's bundled HIP runtime must come up before libfic_hip.so's (capi._torch_first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fic_amd  # noqa: E402
from fic_amd import capi  # noqa: E402

out = {}
for W in (256, 512, 1024):
    r, g, b = (fic_amd.synth.image_u(W, W, 0xC0100 + 3 * W + c).astype(np.int32) for c in range(3))
    argb = ((np.int64(255) << 24) | (r.astype(np.int64) << 16) | (g << 8) | b).astype(np.uint32).view(np.int32).reshape(-1)
    for B in (8,):
        Dw = fic_amd.geometry(W, W, B)[2]
        for wK in (2, 16, Dw):
            capi.encode_rgb(argb, W, W, B, wK)
            reps = 3 if wK == Dw and W >= 1024 else 10
            t0 = time.perf_counter()
            for _ in range(reps):
                capi.encode_rgb(argb, W, W, B, wK)
            dt = (time.perf_counter() - t0) / reps
            nr = (W // B) ** 2
            out[f"{W}x{W}_B{B}_wK{wK}{'_full' if wK == Dw else ''}"] = {"ms_per_call": dt * 1e3, "matches_per_s": nr / dt,
                                                                        "pair_evals_per_s": nr * wK * wK / dt}
print(json.dumps(out, indent=1))

# batched, device-resident contexts (fic_rgb_ctx_*): a batch of colour images, input already in HBM
ctx = {}
for W, planes, wK in ((256, 16, 2), (256, 16, None), (512, 8, 16), (512, 8, None), (1024, 8, None)):
    B = 8
    Dw = fic_amd.geometry(W, W, B)[2]
    wk = Dw if wK is None else wK
    imgs = []
    for p in range(planes):
        r, g, b = (fic_amd.synth.image_u(W, W, 0xC0200 + 7 * p + c).astype(np.int64) for c in range(3))
        imgs.append(((np.int64(255) << 24) | (r << 16) | (g << 8) | b).astype(np.uint32).view(np.int32).reshape(-1))
    t = torch.from_numpy(np.stack(imgs)).cuda()
    with capi.RgbEncoder(W, W, B, wk, planes) as enc:
        enc.set_argb(t)
        s = torch.cuda.current_stream()
        enc.encode(False, s)
        enc.sync()
        reps = 3 if wK is None and W >= 1024 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.encode(False, s)
        enc.sync()
        dt = (time.perf_counter() - t0) / reps
    nr = (W // B) ** 2 * planes
    ctx[f"{planes}x{W}x{W}_B{B}_wK{wk}{'_full' if wK is None else ''}"] = {"ms_per_batch": dt * 1e3, "ms_per_image": dt * 1e3 / planes,
                                                                             "matches_per_s": nr / dt}
print(json.dumps({"rgb_contexts_device_resident": ctx}, indent=1))

# full search: the VALU sweep ("sweep" = 1) beside the matrix-core sweep ("sweep" = 2, the default from 3e7 pairs), device-resident input
cmp = {}
for W, B in ((512, 8), (1024, 8), (1024, 4), (2048, 8), (2048, 16), (4096, 8)):
    Dw = fic_amd.geometry(W, W, B)[2]
    r, g, b = (fic_amd.synth.image_u(W, W, 0xC0300 + c).astype(np.int64) for c in range(3))
    t = torch.from_numpy(((np.int64(255) << 24) | (r << 16) | (g << 8) | b).astype(np.uint32).view(np.int32).reshape(1, -1)).cuda()
    row = {}
    for sweep in (1, 2):
        if sweep == 1 and (W >= 4096 or (B == 16 and W >= 2048)):
            continue                        # seconds per encode on the VALU path (B = 16: the window kernel)
        with capi.RgbEncoder(W, W, B, Dw, 1) as enc:
            enc.set_option("sweep", sweep)
            enc.set_argb(t)
            s = torch.cuda.current_stream()
            enc.encode(False, s)
            enc.sync()
            reps = 2 if W >= 2048 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                enc.encode(False, s)
            enc.sync()
            dt = (time.perf_counter() - t0) / reps
        nr, nd = (W // B) ** 2, Dw * Dw
        row["valu" if sweep == 1 else "matrix_core"] = {"ms_per_image": dt * 1e3, "matches_per_s": nr / dt, "pair_evals_per_s": nr * nd / dt}
    if "valu" in row:
        row["speedup"] = row["valu"]["ms_per_image"] / row["matrix_core"]["ms_per_image"]
    cmp[f"{W}x{W}_B{B}_full"] = row
print(json.dumps({"rgb_full_search_valu_vs_matrix_core": cmp}, indent=1))

# natural-image check: the prune bound E_r of k_sweep_q<NK, 3> scales with the GLOBAL Amax = max_d ||A_d||, so its
# efficiency on natural images is a separate question from the U figure.  LenaColored (tests/golden) tiled to 1024x1024 with a
# per-tile shift, B = 8, full search, beside three U planes of the same size; both sweeps.
lena = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lena_colored_256.npy")).astype(np.int64)
nat = {}
for W in (512, 1024):
    tiles = W // 256
    big = np.zeros((W, W, 3), np.int64)
    for ty in range(tiles):
        for tx in range(tiles):
            big[256 * ty:256 * ty + 256, 256 * tx:256 * tx + 256] = np.roll(lena, (17 * ty + 5 * tx, 11 * tx + 3 * ty), axis=(0, 1))
    lena_argb = ((np.int64(255) << 24) | (big[..., 0] << 16) | (big[..., 1] << 8) | big[..., 2]).astype(np.uint32).view(np.int32).reshape(1, -1)
    r, g, b = (fic_amd.synth.image_u(W, W, 0xC0400 + c).astype(np.int64) for c in range(3))
    u_argb = ((np.int64(255) << 24) | (r << 16) | (g << 8) | b).astype(np.uint32).view(np.int32).reshape(1, -1)
    Dw = fic_amd.geometry(W, W, 8)[2]
    for name, argb in (("LenaColored_tiled", lena_argb), ("U", u_argb)):
        t = torch.from_numpy(argb).cuda()
        row = {}
        for sweep in (1, 2):
            with capi.RgbEncoder(W, W, 8, Dw, 1) as enc:
                enc.set_option("sweep", sweep)
                enc.set_argb(t)
                s = torch.cuda.current_stream()
                enc.encode(False, s)
                enc.sync()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    enc.encode(False, s)
                enc.sync()
                row["valu" if sweep == 1 else "matrix_core"] = {"ms_per_image": (time.perf_counter() - t0) / reps * 1e3}
        row["speedup"] = row["valu"]["ms_per_image"] / row["matrix_core"]["ms_per_image"]
        nat[f"{W}x{W}_B8_full_{name}"] = row
print(json.dumps({"rgb_full_search_natural_vs_U": nat}, indent=1))
