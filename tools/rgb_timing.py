#!/usr/bin/env python3
"""Timing of the joint-RGB encoder (fic_encode_rgb_argb = FractalCompression.encodeRGB, FC:171-219) through the one-shot
C entry point: host ARGB in, host codebook out.  Synthetic colour images: three U planes."""
import json
import os
import sys
import time

import numpy as np

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
