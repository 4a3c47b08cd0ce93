"""Timing + k_sweep_q counters for a list of cases: W,B,iso,planes[,dist[,sweep[,chunks]]] (defaults: U, 6, auto).
dist: U = iid bytes, S = flat tiles + noise, L = LenaGrey (tests/golden, 256x256) tiled with a per-plane shift (exact duplicate
blocks: ties everywhere), N = LenaGrey enlarged bilinearly to W x W and shifted per plane (a smooth natural image, no duplicates)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fic_amd
from fic_amd import synth


_LENA = None


def lena(W, seed, enlarge):
    global _LENA
    if _LENA is None:
        _LENA = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lena_grey_256.npy"))
    oy, ox = (seed * 7) % 256, (seed * 13) % 256
    if not enlarge:
        t = np.tile(_LENA, (W // 256 + 2, W // 256 + 2))
        return np.ascontiguousarray(t[oy:oy + W, ox:ox + W])
    return synth.enlarge(_LENA, W, W, oy, ox)


def run(W, B, n_iso, planes, dist="U", sweep=6, chunks=0, reps=5):
    if dist in ("L", "N"):
        g = np.stack([lena(W, 100 + 3 * p, dist == "N") for p in range(planes)])
    else:
        f = synth.image_u if dist == "U" else synth.image_s
        g = np.stack([f(W, W, 100 + 3 * p) for p in range(planes)])
    d = torch.from_numpy(g).cuda()
    enc = fic_amd.Encoder(W, W, B, None, n_iso, planes)
    enc.set_gray(d)
    enc.set_option("sweep", sweep)
    enc.set_option("time_sweep", 1)
    if chunks:
        enc.set_option("chunks", chunks)
    if sweep == 6:
        enc.set_option("sweep_stats", 1)
    if os.environ.get("FIC_Q_NOFLAG"):
        enc.set_option("q_noflag", 1)
    s = torch.cuda.current_stream()
    for _ in range(2):
        enc.encode(0, -1, s)
    enc.sync()
    enc.sweep_time()
    if sweep == 6:
        enc.sweep_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        enc.encode(0, -1, s)
    e1.record(s)
    enc.sync()
    total = e0.elapsed_time(e1) / reps
    ms, n = enc.sweep_time()
    st = enc.sweep_stats() if sweep == 6 else {}
    info = enc.info()
    evals = planes * enc.n_ranges * enc.n_domains * n_iso
    nk = B * B // 16
    tiles = st.get("tiles", 0) / reps if st else 0
    msg = (f"W={W} B={B} iso={n_iso} planes={planes} dist={dist} sweep={sweep} chunks={info['chunks']}: {ms / n:.3f} ms/sweep "
           f"({total:.3f} ms/encode), {evals / (ms / n * 1e-3):.3e} evals/s, mfma_frac={evals * 2 * B * B / (ms / n * 1e-3) / 2.5e15:.3f}")
    if st:
        msg += (f", cycles/tile@2.4GHz={ms / n * 1e-3 * 2.4e9 * 1024 / max(tiles, 1):.0f} (floor {32 * nk}), "
                f"flagged_tiles={st['flagged_tiles'] / max(st['tiles'], 1):.4f}, "
                f"exact_pairs/range={st['exact_pairs'] / reps / (planes * enc.n_ranges):.1f}, waves={st['waves'] // reps}, "
                f"clock={st['clock_ghz'] or 0:.3f} GHz, wave_alive={st['wave_ticks'] * 1e-8 / max(st['waves_sampled'], 1) / (ms / n * 1e-3):.2f} of sweep")
    print(msg, flush=True)
    enc.close()


if __name__ == "__main__":
    cases = sys.argv[1:] or ["512,8,8,64", "512,8,8,64,U,3", "512,8,1,64", "512,8,8,1", "512,8,8,1,U,3", "2048,4,1,1", "1024,4,8,4",
                             "2048,16,8,1", "4096,8,8,1"]
    for c in cases:
        p = c.split(",")
        run(int(p[0]), int(p[1]), int(p[2]), int(p[3]), p[4] if len(p) > 4 else "U", int(p[5]) if len(p) > 5 else 6,
            int(p[6]) if len(p) > 6 else 0, reps=2 if int(p[0]) >= 4096 else 5)
