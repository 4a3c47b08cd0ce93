"""k_sweep_d4 (fic_d4.hip): the default VALU sweep for n_iso = 8 at B = 8 takes the 8 isometries algebraically (D4 group
Fourier transform of range and domain blocks, 48 v_dot2c per pair instead of 128 v_dot4).  Same bar as every other sweep:
bit-identical codebooks against the oracle, and against k_sweep_fast at sizes the oracle cannot reach."""
import os

import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import GOLDEN, same_f32

pytestmark = pytest.mark.gpu

IMAGES = {
    "lena64": np.load(os.path.join(GOLDEN, "lena64.npy")),
    "lena256": np.load(os.path.join(GOLDEN, "lena_grey_256.npy")),
    "U128": synth.image_u(128, 128, synth.SEEDS["cfg2"]),
    "S128": synth.image_s(128, 128, synth.SEEDS["cfg2"]),
    "S256": synth.image_s(256, 256, synth.SEEDS["cfg3"]),
    "flat64": np.full((64, 64), 77, np.uint8),
    "U200": synth.image_u(200, 200, 11),     # N_r = 625: partial range tile
    "S200": synth.image_s(200, 200, 12),
    "sat64": (np.indices((64, 64)).sum(axis=0) % 2 * 255).astype(np.uint8),   # checkerboard 0/255: largest slot magnitudes
}


def _same(a, b):
    for k in ("idx_local", "iso", "qrows", "idx_global"):
        assert (a[k] == b[k]).all(), k
    for k in ("a", "b", "err"):
        assert same_f32(a[k], b[k]), k


@pytest.mark.parametrize("name", sorted(IMAGES))
@pytest.mark.parametrize("chunks", [0, 1, 3])
def test_d4_sweep_matches_oracle(oracle, name, chunks):
    g = IMAGES[name]
    h, w = g.shape
    Dw = fic_amd.geometry(w, h, 8)[2]
    ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, 8, Dw, 8)
    got = fic_amd.encode_gray(g, 8, None, 8, sweep=5, chunks=chunks)
    assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
    assert (got["iso"] == ref["iso"]).all()
    assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
    assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
    assert same_f32(got["err"], ref["err"])


def test_d4_is_the_valu_only_choice_where_it_exists_and_refused_elsewhere():
    g = IMAGES["U128"]
    with fic_amd.Encoder(256, 256, 8, None, 8) as enc:
        enc.set_gray(IMAGES["lena256"])
        enc.encode()
        assert enc.info()["sweep_kind"] == 6          # library default: the matrix-core sweep
        auto = {k: v.copy() for k, v in enc.results().items()}
        enc.set_option("sweep", 5)
        enc.encode()
        assert enc.info()["sweep_kind"] == 5
        _same(auto, enc.results())
        enc.set_option("sweep", 2)
        enc.encode()
        assert enc.info()["sweep_kind"] == 2
        _same(auto, enc.results())
    for B, n_iso in [(8, 1), (4, 8), (16, 8)]:
        with fic_amd.Encoder(128, 128, B, None, n_iso) as enc:
            enc.set_gray(g)
            enc.encode()
            # below 2e6 (range, domain) pairs a launch stays on the VALU sweep, above it takes the matrix-core sweep
            assert enc.info()["sweep_kind"] == (6 if enc.n_ranges * enc.n_domains >= 2000000 else 2)
            enc.set_option("sweep", 5)
            with pytest.raises(fic_amd.FicError):
                enc.encode()
    with pytest.raises(fic_amd.FicError):
        fic_amd.encode_gray(IMAGES["lena256"], 8, 16, 8, sweep=5)       # windowed search is not a sweep


def test_d4_equals_fast_sweep_batched_and_sharded():
    imgs = np.stack([synth.image_u(512, 512, synth.SEEDS["cfg2"]), synth.image_s(512, 512, 99), synth.image_u(512, 512, 5)])
    res = {}
    for sweep in (2, 5):
        with fic_amd.Encoder(512, 512, 8, None, 8, planes=3) as enc:
            enc.set_option("sweep", sweep)
            enc.set_gray(imgs)
            enc.encode()
            res[sweep] = {k: v.copy() for k, v in enc.results().items()}
    _same(res[2], res[5])
    g = synth.image_u(1024, 1024, 21)
    whole = fic_amd.encode_gray(g, 8, None, 8, sweep=2)
    with fic_amd.Encoder(1024, 1024, 8, None, 8) as enc:
        enc.set_option("sweep", 5)
        enc.set_gray(g)
        parts = []
        for b, c in fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3):
            enc.encode(b, c)
            r = enc.results()
            parts.append({k: v[0][b:b + c].copy() for k, v in r.items()})
    _same({k: np.concatenate([p[k] for p in parts]) for k in parts[0]}, whole)
