#!/usr/bin/env python3
"""PCIe-inclusive rates of the one-shot C entry point (host buffers in, host results out; the working set of a
geometry is cached between calls, `ms_per_call_cold` is the first call after fic_release_cache; run with FIC_SWEEP=3
in the environment for the matrix-core sweeps) and a full-size encode -> decode round trip.  Never bench.py's `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fic_amd  # noqa: E402

out = {}
for name, (W, B, n_iso) in {"cfg2_single_8iso": (512, 8, 8), "cfg2_single_1iso": (512, 8, 1), "1024_8iso": (1024, 8, 8),
                            "cfg3_2048_B4_1iso": (2048, 4, 1), "cfg4_4096_1iso": (4096, 8, 1)}.items():
    g = fic_amd.synth.image_u(W, W, 0xF1C0000 + W)
    from fic_amd import capi
    capi.release_cache()
    capi.encode_gray_oneshot(g, B, None, n_iso)                 # warm (code object load); creates the working set
    capi.release_cache()
    t0 = time.perf_counter()
    capi.encode_gray_oneshot(g, B, None, n_iso)                 # cold: allocates the working set
    cold = time.perf_counter() - t0
    for _ in range(2):
        capi.encode_gray_oneshot(g, B, None, n_iso)             # first uses of a fresh working set still fault pages in
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        r = capi.encode_gray_oneshot(g, B, None, n_iso)         # warm: working set reused from the cache
    dt = (time.perf_counter() - t0) / reps
    nr = (W // B) ** 2
    out[name] = {"ms_per_call_cold": cold * 1e3, "ms_per_call": dt * 1e3, "matches_per_s_pcie_inclusive": nr / dt, "N_r": nr}

# full-size round trip: 4096x4096 S image (piecewise flat + noise), B=8, reference algorithm, GPU encode + GPU decode
W = 4096
g = fic_amd.synth.image_s(W, W, fic_amd.synth.SEEDS["cfg4"])
with fic_amd.Encoder(W, W, 8, None, 1) as enc:
    enc.set_gray(g)
    t0 = time.perf_counter(); enc.encode(); enc.sync(); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); dec, avg, it = enc.decode(); t_dec = time.perf_counter() - t0
d = dec[0].astype(np.float64) - g.astype(np.float64)
mse = float(np.mean(d * d))
out["roundtrip_4096_S_B8_iso1"] = {"encode_ms": t_enc * 1e3, "decode_ms": t_dec * 1e3, "decode_iterations": int(it[0]),
                                   "avgError": float(avg[0]), "psnr_db": 10 * np.log10(255.0 ** 2 / mse) if mse else None}
print(json.dumps(out, indent=1))
