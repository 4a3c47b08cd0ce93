"""Synthetic grey inputs of SURVEY.md section 8(d): integer-only, identical in every language.

  U : pix(x,y) = splitmix64(seed + y*W + x) >> 56              iid uniform bytes (throughput)
  S : 32x32 tiles of levels {0,64,128,192}, alternate tiles carrying 2-bit noise from U --
      produces flat domain blocks (variance 0), flat ranges (rem 0), exact error ties and
      NaN fits: the adversarial cases for parity.
"""
import numpy as np

SEEDS = {"cfg2": 0xF1C0002, "cfg3": 0xF1C0003, "cfg4": 0xF1C0004, "cfg5": 0xF1C0005}
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    """Output function of SplitMix64 applied to state z (uint64 array)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return z ^ (z >> np.uint64(31))


def image_u(w, h, seed):
    idx = np.arange(w * h, dtype=np.uint64)
    with np.errstate(over="ignore"):
        v = splitmix64((idx + np.uint64(seed)) & _M) >> np.uint64(56)
    return v.astype(np.uint8).reshape(h, w)


def image_s(w, h, seed):
    u = image_u(w, h, seed).astype(np.int32)
    y, x = np.mgrid[0:h, 0:w]
    t = (x >> 5) + (y >> 5)
    noisy = (((x >> 5) ^ (y >> 5)) & 1).astype(bool)
    s = ((t & 3) << 6) + np.where(noisy, u >> 6, 0)
    return s.astype(np.uint8)


def enlarge(base, w, h, oy=0, ox=0):
    """`base` (uint8 [H0, W0], continued periodically) enlarged bilinearly to w x h, shifted by (oy, ox) base pixels: a
    smooth natural image of any size from a small one, without the exact duplicate blocks tiling would give."""
    H0, W0 = base.shape
    ys = np.arange(h) * (H0 / float(h)) + oy
    xs = np.arange(w) * (W0 / float(w)) + ox
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    wy, wx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = base.astype(np.float64)
    at = lambda yy, xx: a[np.ix_(yy % H0, xx % W0)]
    v = (1 - wy) * ((1 - wx) * at(y0, x0) + wx * at(y0, x0 + 1)) + wy * ((1 - wx) * at(y0 + 1, x0) + wx * at(y0 + 1, x0 + 1))
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def image(kind, w, h, seed):
    return image_u(w, h, seed) if kind.upper() == "U" else image_s(w, h, seed)


def images_u_torch(w, h, seeds, device="cuda"):
    """The U images of `seeds` as one uint8 torch tensor [len(seeds), h, w], generated on `device` (the same integers as
    image_u: SplitMix64 in wrapping int64 arithmetic with logical shifts).  bench.py builds its batches with this: the
    numpy generator needs seconds per 16 M pixels on the host."""
    import torch

    def s64(c):                       # uint64 constant as the int64 with the same bits
        return c - (1 << 64) if c >= (1 << 63) else c

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    idx = torch.arange(w * h, dtype=torch.int64, device=device)
    out = torch.empty((len(seeds), h, w), dtype=torch.uint8, device=device)
    for i, seed in enumerate(seeds):
        z = idx + s64(int(seed) & 0xFFFFFFFFFFFFFFFF)
        z = z + s64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        out[i] = lsr(z, 56).to(torch.uint8).view(h, w)
    return out
