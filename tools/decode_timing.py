"""Decoder timing: 4096x4096 S image, B=8, reference algorithm (n_iso=1): GPU encode, then the GPU decoder (FC:356-421) -- wall time of
fic_ctx_decode_host (includes the 16.8 MB copy of the decoded image to the host) and of the device loop alone."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fic_amd
from fic_amd import synth

out = {}
CASES = {"4096_S": (4096, "S"), "4096_U": (4096, "U"), "2048_S": (2048, "S")}
for name in (sys.argv[1:] or list(CASES)):
    size, kind = CASES[name]
    g = synth.image(kind, size, size, synth.SEEDS["cfg4"])
    with fic_amd.Encoder(size, size, 8, None, 1) as enc:
        enc.set_gray(g)
        enc.encode(); enc.sync()
        enc.decode()                       # warm: allocations, code object
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); dec, avg, it = enc.decode(); ts.append(time.perf_counter() - t0)
        out[name] = {"decode_ms_min": min(ts) * 1e3, "decode_ms_median": sorted(ts)[2] * 1e3, "iterations": int(it[0]), "avgError": float(avg[0])}
print(json.dumps(out))
