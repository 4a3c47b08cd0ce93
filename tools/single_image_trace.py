"""The literal BASELINE config 2 -- ONE 512x512 image, B = 8, full search, 8 isometries -- as 300 back-to-back encodes on one stream:
run under `rocprofv3 --kernel-trace --stats` to see what the kernels of an encode (two for a launch this small: k_prep_q8 and the sweep with its finalising tail) cost on the GPU against the time per encode
(what is left is dispatch gaps between dependent kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fic_amd
from fic_amd import synth

n_iso = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dist = sys.argv[2] if len(sys.argv) > 2 else "U"          # U = iid bytes, N = LenaGrey (tests/golden) enlarged bilinearly
if dist == "N":
    import numpy as np
    g = synth.enlarge(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lena_grey_256.npy")), 512, 512)
else:
    g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
enc = fic_amd.Encoder(512, 512, 8, None, n_iso)
enc.set_gray(torch.from_numpy(g[None]).cuda())
s = torch.cuda.Stream()
for _ in range(20):
    enc.encode(0, -1, s)
s.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(300):
    enc.encode(0, -1, s)
e1.record(s)
s.synchronize()
print(json.dumps({"n_iso": n_iso, "dist": dist, "ms_per_encode": e0.elapsed_time(e1) / 300, "kernel": enc.last_kernel(), "chunks": enc.info()["chunks"]}))
enc.close()
