"""Joint-RGB full search on the matrix cores (k_sweep_q<NK, 3>, option "sweep" = 2 of fic_rgb_ctx): bit-identical to the
oracle's encodeRGB (FC:171-219, 697-735, 760-808) and to the VALU sweeps ("sweep" = 1) on the same inputs."""
import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import same_f32

pytestmark = pytest.mark.gpu


def _rgb_synth(w, h, seed, flat=False):
    f = synth.image_s if flat else synth.image_u
    return np.stack([f(w, h, seed), f(w, h, seed + 1), f(w, h, seed + 2)], axis=-1)


def _encode(argbs, w, h, B, sweep, chunks=0):
    Dw = fic_amd.geometry(w, h, B)[2]
    with fic_amd.capi.RgbEncoder(w, h, B, Dw, planes=len(argbs)) as enc:
        enc.set_option("sweep", sweep)
        enc.set_option("chunks", chunks)
        enc.set_argb(np.stack(argbs))
        enc.encode(with_collage=False)
        r = enc.results()
        assert enc.last_sweep() == (2 if sweep == 2 else 1)
    return r


def _same(a, b, p=None):
    for k in ("idx_local", "qrows"):
        x = a[k] if p is None else a[k][p]
        assert (x == b[k]).all(), k
    for k in ("a", "bR", "bG", "bB"):
        x = a[k] if p is None else a[k][p]
        assert same_f32(x, b[k]), k


def _oracle_dict(oracle, argb, w, h, B):
    Dw = fic_amd.geometry(w, h, B)[2]
    ref = oracle.encode_rgb(argb, w, h, B, Dw)
    return {"idx_local": ref[:, 0].astype(np.int32), "a": ref[:, 1], "bR": ref[:, 2], "bG": ref[:, 3], "bB": ref[:, 4],
            "qrows": oracle.quantise_rgb(ref)}


@pytest.mark.parametrize("size,B,flat", [(64, 4, False), (64, 4, True), (128, 8, False), (128, 8, True), (200, 8, False),
                                         (200, 4, True), (256, 16, False), (256, 16, True), (192, 8, True)])
def test_matrix_core_full_search_matches_the_oracle(oracle, size, B, flat):
    """U planes (no ties) and S planes (flat blocks, varianzDomain == 0, exact error ties, 0/0 fits); N_r not a multiple
    of 32 (200x200), all three block sizes (B = 16 has no VALU full-search kernel: it was the window kernel before)."""
    rgb = _rgb_synth(size, size, 900 + B + size, flat)
    argb = oracle.rgb_to_argb(rgb)
    ref = _oracle_dict(oracle, argb, size, size, B)
    _same(_encode([argb], size, size, B, 2), ref, 0)
    for chunks in (3, 40, 100000):          # up to one chunk per domain tile: every tile seeds theta out of order, theta_g is shared
        _same(_encode([argb], size, size, B, 2, chunks), ref, 0)


def test_lena_colored_and_extreme_images(oracle, lena_colored):
    for B in (4, 8, 16):
        argb = oracle.rgb_to_argb(np.ascontiguousarray(lena_colored[:128, :128]))
        _same(_encode([argb], 128, 128, B, 2), _oracle_dict(oracle, argb, 128, 128, B), 0)
    sat = np.zeros((64, 64, 3), np.uint8)
    sat[::2, :, 0] = 255
    sat[:, ::2, 1] = 255
    sat[32:, :, 2] = 255                                          # the largest |greyR * greyD| sums: f32 accumulation order
    const = np.full((64, 64, 3), (10, 200, 30), np.uint8)        # constant colour: varianzRange == 0 everywhere
    grad = np.zeros((64, 64, 3), np.uint8)
    grad[..., 0] = np.arange(64)[None, :] * 4
    grad[..., 1] = np.arange(64)[:, None] * 4
    grad[..., 2] = 128                                            # smooth ramps: many near-equal candidates
    for img in (sat, const, grad):
        for B in (4, 8):
            argb = oracle.rgb_to_argb(img)
            _same(_encode([argb], 64, 64, B, 2), _oracle_dict(oracle, argb, 64, 64, B), 0)


@pytest.mark.parametrize("seed", range(6))
def test_low_depth_colour_images_with_exact_ties(oracle, seed):
    """1- and 2-bit channels, blocky structure: many candidates share the best error EXACTLY (the strict '<' of FC:707 keeps
    the lowest index), varianzDomain and varianzRange hit 0 often, kovarianz sums cancel -- the any-order rules of the
    prune (seeding from a later pair, theta published across pool chunks) must never drop the earlier of two equal pairs."""
    rng = np.random.default_rng(1234 + seed)
    size, B = (64, 4) if seed % 2 == 0 else (128, 8)
    bits = 1 + seed % 2
    cell = (2, 4, 8)[seed % 3]
    low = rng.integers(0, 1 << bits, (size // cell, size // cell, 3), dtype=np.uint8)
    rgb = np.repeat(np.repeat(low, cell, 0), cell, 1) * (255 // ((1 << bits) - 1))
    if seed >= 3:                                                  # a few isolated pixels break the block symmetry
        ys, xs = rng.integers(0, size, 40), rng.integers(0, size, 40)
        rgb[ys, xs] = rng.integers(0, 256, (40, 3), dtype=np.uint8)
    argb = oracle.rgb_to_argb(np.ascontiguousarray(rgb.astype(np.uint8)))
    ref = _oracle_dict(oracle, argb, size, size, B)
    for chunks in (0, 5, 100000):
        _same(_encode([argb], size, size, B, 2, chunks), ref, 0)
    _same(_encode([argb], size, size, B, 1), ref, 0)


@pytest.mark.parametrize("size,B,flat", [(512, 8, False), (512, 8, True), (512, 4, False), (1024, 16, False), (768, 8, True)])
def test_matrix_core_equals_valu_sweep_on_large_images(oracle, size, B, flat):
    """Sizes the scalar oracle does not finish in seconds: the two GPU sweeps (different arithmetic paths to the same
    reference arithmetic) must give the same codebook; a sample of range rows is also checked against the oracle."""
    rgb = _rgb_synth(size, size, 4100 + B, flat)
    argb = oracle.rgb_to_argb(rgb)
    q = _encode([argb], size, size, B, 2)
    v = _encode([argb], size, size, B, 1)
    for k in ("idx_local", "qrows"):
        assert (q[k] == v[k]).all(), k
    for k in ("a", "bR", "bG", "bB"):
        assert same_f32(q[k], v[k]), k
    assert flat or len(np.unique(q["idx_local"])) > 10


def test_default_selection_and_batches(oracle, lena_colored):
    """Automatic choice: 512x512 B=8 full search (6.4e7 pairs) runs the matrix-core sweep, a 128x128 one the VALU sweep;
    a batch of three images in one context equals the per-image results."""
    imgs = [np.ascontiguousarray(lena_colored[:128, :128]), _rgb_synth(128, 128, 41), _rgb_synth(128, 128, 52, flat=True)]
    argbs = [oracle.rgb_to_argb(x) for x in imgs]
    Dw = fic_amd.geometry(128, 128, 8)[2]
    with fic_amd.capi.RgbEncoder(128, 128, 8, Dw, planes=3) as enc:
        enc.set_argb(np.stack(argbs))
        enc.encode()
        assert enc.last_sweep() == 1
        enc.set_option("sweep", 2)
        enc.encode()
        assert enc.last_sweep() == 2
        r = enc.results()
    for p, argb in enumerate(argbs):
        _same(r, _oracle_dict(oracle, argb, 128, 128, 8), p)
    big = oracle.rgb_to_argb(_rgb_synth(512, 512, 7))
    Dw = fic_amd.geometry(512, 512, 8)[2]
    with fic_amd.capi.RgbEncoder(512, 512, 8, Dw) as enc:
        enc.set_argb(big[None])
        enc.encode()
        assert enc.last_sweep() == 2
        with pytest.raises(fic_amd.FicError):
            enc.set_option("sweep", 3)
    fic_amd.capi.release_cache()


def test_seeded_fuzz_rgb_full_search(oracle):
    """Random (size, B, image kind, chunk count) colour cases: both full-search sweeps against the oracle's encodeRGB."""
    import os
    rng = np.random.default_rng(int(os.environ.get("FIC_FUZZ_SEED", "20261004")))
    for _ in range(int(os.environ.get("FIC_FUZZ_CASES", "40"))):
        B = int(rng.choice([4, 8, 16]))
        R = int(rng.integers(2, 144 // B + 1))
        size = R * B
        kind = rng.choice(["U", "S", "low", "ramp", "const"])
        seed = int(rng.integers(1, 1 << 30))
        if kind in ("U", "S"):
            rgb = _rgb_synth(size, size, seed, flat=(kind == "S"))
        elif kind == "low":
            bits, cell = int(rng.integers(1, 4)), int(rng.choice([1, 2, 4]))
            n = -(-size // cell)
            low = rng.integers(0, 1 << bits, (n, n, 3), dtype=np.uint8)
            rgb = (np.repeat(np.repeat(low, cell, 0), cell, 1)[:size, :size] * (255 // ((1 << bits) - 1))).astype(np.uint8)
        elif kind == "ramp":
            y, x = np.mgrid[0:size, 0:size]
            rgb = np.stack([(x * 3 + seed) % 256, (y * 5 + x) % 256, (x + y + seed) % 256], -1).astype(np.uint8)
        else:
            rgb = np.full((size, size, 3), (seed % 256, (seed >> 8) % 256, (seed >> 16) % 256), np.uint8)
        argb = oracle.rgb_to_argb(np.ascontiguousarray(rgb))
        ref = _oracle_dict(oracle, argb, size, size, B)
        for sweep in (1, 2):
            chunks = int(rng.integers(0, 6)) if sweep == 2 else 0
            try:
                _same(_encode([argb], size, size, B, sweep, chunks), ref, 0)
            except AssertionError as e:
                raise AssertionError(f"case size={size} B={B} kind={kind} seed={seed} sweep={sweep} chunks={chunks}: {e}")
