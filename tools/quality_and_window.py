#!/usr/bin/env python3
"""(a) decoded PSNR of LenaGrey.png: reference algorithm (n_iso=1) vs the 8-isometry extension, GPU encode + GPU decode;
(b) run time of the GUI's windowed searches (k_sweep_generic) at 4096x4096."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fic_amd  # noqa: E402

out = {"psnr_lena256": {}, "window_4096": {}}
g = np.load(os.path.join(ROOT, "tests", "golden", "lena_grey_256.npy"))
for B in (4, 8, 16):
    for n_iso in (1, 8):
        with fic_amd.Encoder(256, 256, B, None, n_iso) as enc:
            enc.set_gray(g)
            enc.encode()
            dec, avg, it = enc.decode()
        d = dec[0].astype(np.float64) - g
        out["psnr_lena256"][f"B{B}_iso{n_iso}_full"] = {"psnr_db": round(float(10 * np.log10(255.0 ** 2 / np.mean(d * d))), 3),
                                                       "decode_iterations": int(it[0])}
big = fic_amd.synth.image_s(4096, 4096, 77)
for B, wK in ((8, 2), (8, 16), (4, 16), (16, 16)):
    with fic_amd.Encoder(4096, 4096, B, wK, 1) as enc:
        enc.set_gray(big)
        enc.set_option("time_sweep", 1)
        enc.encode(); enc.sync()
        enc.sweep_time()
        t0 = time.perf_counter()
        for _ in range(3):
            enc.encode()
        enc.sync()
        wall = (time.perf_counter() - t0) / 3
        ms, n = enc.sweep_time()
    out["window_4096"][f"B{B}_wK{wK}"] = {"encode_ms": round(wall * 1e3, 3), "sweep_ms": round(ms / n, 3),
                                         "matches_per_s": round((4096 // B) ** 2 / wall)}
print(json.dumps(out, indent=1))
