"""Diagnostic (a build with -DFIC_Q_PHASES): where a wave of the single-image sweep spends its cycles -- prologue | loop | final flush."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import fic_amd
from fic_amd import synth, capi
for n_iso, chunks in ((8, 0), (8, 8), (1, 0)):
    g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
    enc = fic_amd.Encoder(512, 512, 8, None, n_iso)
    enc.set_gray(torch.from_numpy(g[None]).cuda())
    enc.set_option("sweep_stats", 1)
    enc.set_option("time_sweep", 1)
    if chunks:
        enc.set_option("chunks", chunks)
    for _ in range(5):
        enc.encode()
    enc.sync(); enc.sweep_time()
    v = (C.c_uint64 * 8)(); capi.check(capi.lib().fic_ctx_sweep_stats(enc._h, v, 1))
    reps = 50
    for _ in range(reps):
        enc.encode()
    enc.sync()
    ms, n = enc.sweep_time()
    capi.check(capi.lib().fic_ctx_sweep_stats(enc._h, v, 1))
    v = [int(x) for x in v]
    w = max(v[6], 1)
    ghz = v[4] / v[5] / 10.0 if v[5] else 0
    us = lambda c: c / w / (ghz * 1e3)
    print(f"n_iso={n_iso} chunks={enc.info()['chunks']}: sweep {ms / n * 1e3:.1f} us; per sampled wave: prologue {us(v[0]):.1f} us, loop {us(v[1]):.1f} us, final flush {us(v[2]):.1f} us, "
          f"alive {us(v[4]):.1f} us at {ghz:.2f} GHz; slow tiles/wave {v[3] / w:.1f}, queued pairs/wave {v[7] / w:.1f}")
    enc.close()
