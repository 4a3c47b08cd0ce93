#!/usr/bin/env python3
"""Regenerates the data fixtures under tests/golden/ from the reference's own data files.

Run in the build container only (needs /root/reference and PIL):
    python tests/golden/make_fixtures.py

Fixtures are DATA (decoded pixels of the reference's input images and its committed
encoder output), not source:
  lena_grey_256.npy    uint8 [256,256]   R channel of LenaGrey.png (all r=g=b)
  lena64.npy           uint8 [64,64]     R channel of Lena64.png   (all r=g=b, alpha 255)
  lena_colored_256.npy uint8 [256,256,3] LenaColored.jpg decoded by libjpeg (PIL)
  unknown_run.bin      the reference's committed encoder output unknown.run (K1)
  k2_animation_gif.json the five "MSE" labels read off Animation.gif frames (K2)
"""
import hashlib
import json
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    g = np.asarray(Image.open(os.path.join(REF, "LenaGrey.png")).convert("RGB"))
    assert g.shape == (256, 256, 3) and (g[..., 0] == g[..., 1]).all() and (g[..., 1] == g[..., 2]).all()
    np.save(os.path.join(HERE, "lena_grey_256.npy"), np.ascontiguousarray(g[..., 0]))

    im = Image.open(os.path.join(REF, "Lena64.png")).convert("RGBA")
    s = np.asarray(im)
    assert s.shape == (64, 64, 4) and (s[..., 0] == s[..., 1]).all() and (s[..., 1] == s[..., 2]).all()
    assert (s[..., 3] == 255).all()
    np.save(os.path.join(HERE, "lena64.npy"), np.ascontiguousarray(s[..., 0]))

    c = np.asarray(Image.open(os.path.join(REF, "LenaColored.jpg")).convert("RGB"))
    assert c.shape == (256, 256, 3)
    np.save(os.path.join(HERE, "lena_colored_256.npy"), np.ascontiguousarray(c))

    shutil.copyfile(os.path.join(REF, "unknown.run"), os.path.join(HERE, "unknown_run.bin"))
    os.chmod(os.path.join(HERE, "unknown_run.bin"), 0o644)

    # K2: GUI "MSE" label = FractalCompression.avgError after decode (RLEAppController.java:180),
    # read off Animation.gif (README.md:7).  Strings exactly as Java's Float.toString printed them.
    k2 = [
        {"gif_frame": 0, "B": 16, "wK": 16, "mse_label": "0.3744049", "ssd": 24537},
        {"gif_frame": 11, "B": 8, "wK": 16, "mse_label": "0.3647766", "ssd": 23906},
        {"gif_frame": 31, "B": 4, "wK": 16, "mse_label": "0.73760986", "ssd": 48340},
        {"gif_frame": 59, "B": 8, "wK": 8, "mse_label": "0.52404785", "ssd": 34344},
        {"gif_frame": 72, "B": 8, "wK": 4, "mse_label": "0.36376953", "ssd": 23840},
    ]
    with open(os.path.join(HERE, "k2_animation_gif.json"), "w") as f:
        json.dump({"image": "lena_grey_256.npy", "cases": k2}, f, indent=1)

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith((".npy", ".bin", ".json")):
            with open(os.path.join(HERE, fn), "rb") as f:
                print(fn, hashlib.sha256(f.read()).hexdigest()[:16])


if __name__ == "__main__":
    main()
