"""BASELINE.json's full sizes (configs 3 and 4: 262 144 range blocks x 1 042 441 domain blocks).
The oracle cannot sweep these, so parity is checked (a) on a handful of range blocks against the
oracle over the FULL pool, and (b) through size-independent properties: the result of a range span
does not depend on how the pool is chunked, which sweep kernel ran, or how ranges are sharded."""
import numpy as np
import pytest

import fic_amd
from fic_amd import capi, synth
from conftest import same_f32

pytestmark = pytest.mark.gpu
XCHECK = capi.has_xcheck()     # round 1's exact-covariance matrix-core sweeps ("sweep" = 3) are in the library (default build)


def _span(enc, b, c):
    enc.encode(b, c)
    r = enc.results()
    return {k: v[0][b:b + c].copy() for k, v in r.items()}


def _same(a, b):
    for k in ("idx_local", "iso", "qrows", "idx_global"):
        assert (a[k] == b[k]).all(), k
    for k in ("a", "b", "err"):
        assert same_f32(a[k], b[k]), k


def _oracle_rows(oracle, g, B, wK, n_iso, rows):
    argb = oracle.gray_to_argb(g)
    h, w = g.shape
    out = {}
    for j in rows:
        e = oracle.encode_gray(argb, w, h, B, wK, n_iso, j, j + 1)
        out[j] = (int(e["info"][j, 0]), e["info"][j, 1:].copy(), int(e["iso"][j]), e["err"][j])
    return out


def test_cfg3_size_2048_B4(oracle):
    g = synth.image_u(2048, 2048, synth.SEEDS["cfg3"])
    with fic_amd.Encoder(2048, 2048, 4, None, 1) as enc:
        assert (enc.n_ranges, enc.n_domains, enc.wK) == (262144, 1042441, 1021)
        enc.set_gray(g)
        b, c = 131072 - 256, 1024                      # a span in the middle of the image, tile aligned
        enc.set_option("sweep", 2)
        base = _span(enc, b, c)
        for chunks in (1, 5, 64):
            enc.set_option("chunks", chunks)
            _same(_span(enc, b, c), base)
        enc.set_option("chunks", 0)
        parts = [_span(enc, b + o, n) for o, n in ((0, 256), (256, 512), (768, 256))]   # shards of the span
        _same({k: np.concatenate([p[k] for p in parts]) for k in base}, base)
        enc.set_option("sweep", 1)                     # generic exact kernel on a few ranges
        _same(_span(enc, b, 8), {k: v[:8] for k, v in base.items()})
    ref = _oracle_rows(oracle, g, 4, 1021, 1, [b, b + 517, b + 1023])
    for j, (idx, ab, iso, err) in ref.items():
        o = j - b
        assert base["idx_local"][o] == idx and same_f32(np.array([base["a"][o], base["b"][o]]), ab)
        assert same_f32(np.array([base["err"][o]]), np.array([err]))


def test_cfg4_size_4096_B8_iso8(oracle):
    g = synth.image_u(4096, 4096, synth.SEEDS["cfg4"])
    with fic_amd.Encoder(4096, 4096, 8, None, 8) as enc:
        assert (enc.n_ranges, enc.n_domains, enc.wK) == (262144, 1042441, 1021)
        enc.set_gray(g)
        b, c = 200000 - 200000 % 64, 512
        enc.set_option("sweep", 2)
        valu = _span(enc, b, c)
        enc.set_option("chunks", 3)
        _same(_span(enc, b, c), valu)
        enc.set_option("chunks", 0)
        enc.set_option("sweep", 3 if XCHECK else 6)    # another matrix-core sweep, same span
        _same(_span(enc, b, c), valu)
        enc.set_option("chunks", 7)
        _same(_span(enc, b, c), valu)
    ref = _oracle_rows(oracle, g, 8, 1021, 8, [b + 3, b + 400])
    for j, (idx, ab, iso, err) in ref.items():
        o = j - b
        assert valu["idx_local"][o] == idx and valu["iso"][o] == iso
        assert same_f32(np.array([valu["a"][o], valu["b"][o]]), ab)


def test_cfg4_size_reference_algorithm_and_S_image(oracle):
    """n_iso = 1 (the reference algorithm) on the adversarial S image at 4096x4096: flat blocks, rem == 0
    ranges and exact ties at scale; the generic exact kernel must agree with the fast one."""
    g = synth.image_s(4096, 4096, synth.SEEDS["cfg4"])
    with fic_amd.Encoder(4096, 4096, 8, None, 1) as enc:
        enc.set_gray(g)
        b, c = 100000 - 100000 % 128, 512
        enc.set_option("sweep", 2)
        fast = _span(enc, b, c)
        enc.set_option("sweep", 1)
        _same(_span(enc, b, 16), {k: v[:16] for k, v in fast.items()})
    ref = _oracle_rows(oracle, g, 8, 1021, 1, [b, b + 300])
    for j, (idx, ab, iso, err) in ref.items():
        o = j - b
        assert fast["idx_local"][o] == idx and same_f32(np.array([fast["a"][o], fast["b"][o]]), ab)


@pytest.mark.parametrize("size,B,n_iso,dist", [(4096, 8, 8, "U"), (4096, 8, 1, "S"), (2048, 4, 1, "U"), (2048, 4, 8, "S"),
                                                (4096, 16, 1, "S"), (2048, 16, 8, "U"), (2048, 8, 8, "lena"), (2048, 8, 1, "lena"),
                                                (1024, 4, 8, "lena"), (2048, 8, 8, "nat"), (2048, 4, 1, "nat"), (1024, 16, 8, "nat")])
def test_whole_codebook_at_full_size_valu_equals_matrix_core(size, B, n_iso, dist):
    """Every range block of a BASELINE-sized image through two independent sweeps -- the VALU kernel (integer dot
    products, range blocks in lanes) and the matrix-core kernel (bf16 / i8 MFMA tiles, deferred exact epilogue): the
    whole codebook, the unquantised fit and the winning errors must be the same bits, and so must the `.run` bytes."""
    import hashlib
    if dist == "lena":                                  # natural-image statistics: LenaGrey tiled with a shift
        import os
        from conftest import GOLDEN
        base = np.load(os.path.join(GOLDEN, "lena_grey_256.npy"))
        g = np.ascontiguousarray(np.tile(base, (size // 256 + 1, size // 256 + 1))[37:37 + size, 101:101 + size])
    elif dist == "nat":                                 # LenaGrey enlarged bilinearly: smooth blocks, thousands of near-equal candidates,
        import os                                       # best L of many range blocks below 0.26 n
        from conftest import GOLDEN
        g = synth.enlarge(np.load(os.path.join(GOLDEN, "lena_grey_256.npy")), size, size, 11, 5)
    else:
        g = (synth.image_u if dist == "U" else synth.image_s)(size, size, synth.SEEDS["cfg4"] + B + n_iso)
    res = {}
    with fic_amd.Encoder(size, size, B, None, n_iso) as enc:
        enc.set_gray(g)
        for sweep in (2, 6) + ((3,) if XCHECK else ()) + ((5,) if (B == 8 and n_iso == 8) else ()):     # 6: k_sweep_q, the default; 5: k_sweep_d4; 3: round 1's
            enc.set_option("sweep", sweep)
            enc.encode()
            res[sweep] = {k: v[0].copy() for k, v in enc.results().items()}
            assert enc.info()["sweep_kind"] == sweep
        wK = enc.wK
    _same(res[2], res[3])
    _same(res[2], res[6])
    if 5 in res:
        _same(res[2], res[5])
    runs = [hashlib.sha256(fic_amd.write_run_gray(res[s]["qrows"], size, size, B, wK)).hexdigest() for s in (2, 3 if XCHECK else 6)]
    assert runs[0] == runs[1]
    # not a degenerate comparison: the codebook uses many different domain blocks (and isometries); the S images are
    # mostly flat 32x32 tiles (rem == 0 -> index 0, FC:677), so only their noisy half spreads out
    assert len(np.unique(res[2]["idx_local"])) > {"S": 10, "lena": 300, "nat": 100, "U": 1000}[dist]   # (tiled Lena repeats itself)
    if n_iso == 8 and dist == "U":
        assert len(np.unique(res[2]["iso"])) == 8


def test_beyond_the_named_sizes_8192_B8(oracle):
    """Four times config 4's pool: 8192 x 8192, B = 8 -- 1 048 576 range blocks x 4 182 025 domain blocks (22 bits of the 24 that
    k_sweep_q's queue entries carry).  A range span through the default sweep: independent of the chunk count, equal to the
    VALU sweep, and three of its rows equal to the oracle's scan of the FULL pool."""
    g = synth.image_u(8192, 8192, 0xF1C0008)
    with fic_amd.Encoder(8192, 8192, 8, None, 1) as enc:
        assert (enc.n_ranges, enc.n_domains, enc.wK) == (1048576, 4182025, 2045)
        enc.set_gray(g)
        b, c = 524288 - 128, 512
        base = _span(enc, b, c)
        assert enc.info()["sweep_kind"] == 6
        for chunks in (1, 9):
            enc.set_option("chunks", chunks)
            _same(_span(enc, b, c), base)
        enc.set_option("chunks", 0)
        enc.set_option("sweep", 2)
        _same(_span(enc, b, 128), {k: v[:128] for k, v in base.items()})
    ref = _oracle_rows(oracle, g, 8, 2045, 1, [b, b + 255, b + 511])
    for j, (idx, ab, iso, err) in ref.items():
        o = j - b
        assert base["idx_local"][o] == idx and same_f32(np.array([base["a"][o], base["b"][o]]), ab)
        assert same_f32(np.array([base["err"][o]]), np.array([err]))
