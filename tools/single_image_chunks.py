"""One 512x512 (or given) image: ms per encode and per sweep for a list of pool chunk counts, WITHOUT the kernel's stats counters
(tools/q_stats.py's per-wave atomics inflate small launches in proportion to their wave count)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fic_amd
from fic_amd import synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dist = sys.argv[2] if len(sys.argv) > 2 else "U"          # U = iid bytes, N = LenaGrey (tests/golden) enlarged bilinearly
out = {}
for n_iso in (8, 1):
    if dist == "N":
        import numpy as np
        g = synth.enlarge(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lena_grey_256.npy")), W, W)
    else:
        g = synth.image_u(W, W, synth.SEEDS["cfg2"])
    d = torch.from_numpy(g[None]).cuda()
    for chunks in (0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 48):
        enc = fic_amd.Encoder(W, W, 8, None, n_iso)
        enc.set_gray(d)
        enc.set_option("sweep", 6)
        enc.set_option("time_sweep", 1)
        if chunks:
            enc.set_option("chunks", chunks)
        s = torch.cuda.Stream()
        for _ in range(10):
            enc.encode(0, -1, s)
        s.synchronize()
        enc.sweep_time()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(200):
            enc.encode(0, -1, s)
        e1.record(s)
        s.synchronize()
        ms, n = enc.sweep_time()
        out[f"{W}_{dist}_iso{n_iso}_chunks{chunks}"] = {"chunks_used": enc.info()["chunks"], "ms_per_encode": round(e0.elapsed_time(e1) / 200, 4), "sweep_ms": round(ms / n, 4)}
        enc.close()
print(json.dumps(out, indent=0))
