"""The opt-in matrix-core sweeps ("sweep" = 3: bf16 operands at B = 4/8 -- fic_bf16.hip -- and i8 operands at
B = 16; "sweep" = 4: i8 operands at every block size -- fic_mfma.hip) are held to the same bar as the default VALU
sweep: bit-identical codebooks against the oracle, and against k_sweep_fast at sizes the oracle cannot reach."""
import os

import numpy as np
import pytest

import fic_amd
from fic_amd import capi
from fic_amd import synth
from conftest import GOLDEN, same_f32

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not capi.has_xcheck(), reason="library built with FIC_BUILD_XCHECK=0: no sweep 3 / 4")]

IMAGES = {
    "lena64": np.load(os.path.join(GOLDEN, "lena64.npy")),
    "lena256": np.load(os.path.join(GOLDEN, "lena_grey_256.npy")),
    "U128": synth.image_u(128, 128, synth.SEEDS["cfg2"]),
    "S128": synth.image_s(128, 128, synth.SEEDS["cfg2"]),
    "S256": synth.image_s(256, 256, synth.SEEDS["cfg3"]),
    "U256": synth.image_u(256, 256, synth.SEEDS["cfg3"]),
    "flat64": np.full((64, 64), 77, np.uint8),
    "U200": synth.image_u(200, 200, 11),     # N_r = 625: partial range group, N_d = 2209: partial domain tile
    "S200": synth.image_s(200, 200, 12),
}


def _same(a, b):
    for k in ("idx_local", "iso", "qrows", "idx_global"):
        assert (a[k] == b[k]).all(), k
    for k in ("a", "b", "err"):
        assert same_f32(a[k], b[k]), k


@pytest.mark.parametrize("name", sorted(IMAGES))
@pytest.mark.parametrize("chunks", [0, 1, 3])
def test_mfma_sweep_matches_oracle(oracle, name, chunks):
    g = IMAGES[name]
    h, w = g.shape
    Dw = fic_amd.geometry(w, h, 8)[2]
    ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, 8, Dw, 8)
    got = fic_amd.encode_gray(g, 8, None, 8, sweep=3, chunks=chunks)
    assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
    assert (got["iso"] == ref["iso"]).all()
    assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
    assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
    assert same_f32(got["err"], ref["err"])


def test_mfma_equals_valu_sweep_at_cfg2_size_and_batched():
    imgs = np.stack([synth.image_u(512, 512, synth.SEEDS["cfg2"]), synth.image_s(512, 512, 99),
                     synth.image_u(512, 512, 5)])
    res = {}
    for sweep in (2, 3):
        with fic_amd.Encoder(512, 512, 8, None, 8, planes=3) as enc:
            enc.set_option("sweep", sweep)
            enc.set_gray(imgs)
            enc.encode()
            res[sweep] = enc.results()
            assert enc.info()["sweep_kind"] == sweep
    _same(res[2], res[3])


def test_mfma_shards_and_1024():
    g = synth.image_u(1024, 1024, 21)
    whole = fic_amd.encode_gray(g, 8, None, 8, sweep=2)
    with fic_amd.Encoder(1024, 1024, 8, None, 8) as enc:
        enc.set_option("sweep", 3)
        enc.set_gray(g)
        parts = []
        for b, c in fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3):
            enc.encode(b, c)
            r = enc.results()
            parts.append({k: v[0][b:b + c].copy() for k, v in r.items()})
    cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    _same(cat, whole)


def test_mfma_is_refused_outside_its_configuration():
    with pytest.raises(fic_amd.FicError):
        fic_amd.encode_gray(IMAGES["lena256"], 8, 16, 1, sweep=3)       # windowed search is not a sweep


# ---- n_iso = 1: the reference algorithm on the matrix cores (k_sweep_mfma1, B = 4 / 8 / 16) -------------------

ISO1 = [(n, B) for n in sorted(IMAGES) for B in (4, 8, 16)
        if not (n in ("lena64", "flat64", "U200", "S200") and B == 16) and not (n in ("lena256", "S256", "U256") and B == 4)]


@pytest.mark.parametrize("name,B", ISO1 + [("lena256", 4)])
def test_mfma1_reference_algorithm_matches_oracle(oracle, name, B):
    g = IMAGES[name]
    h, w = g.shape
    Dw = fic_amd.geometry(w, h, B)[2]
    ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, Dw, 1)
    for chunks in (0, 2):
        got = fic_amd.encode_gray(g, B, None, 1, sweep=3, chunks=chunks)
        assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
        assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
        assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
        assert same_f32(got["err"], ref["err"])


@pytest.mark.parametrize("B,size", [(8, 1024), (4, 512), (16, 1024)])
def test_mfma1_equals_valu_sweep_at_scale_and_sharded(B, size):
    imgs = np.stack([synth.image_u(size, size, 31 + B), synth.image_s(size, size, 32 + B)])
    with fic_amd.Encoder(size, size, B, None, 1, planes=2) as enc:
        enc.set_gray(imgs)
        enc.set_option("sweep", 2)
        enc.encode()
        valu = enc.results()
        enc.set_option("sweep", 3)
        spans = fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3)
        parts = []
        for b, c in spans:
            enc.encode(b, c)
            r = enc.results()
            parts.append({k: v[:, b:b + c].copy() for k, v in r.items()})
        assert enc.info()["sweep_kind"] == 3
    cat = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
    _same(cat, valu)


def test_fic_sweep_environment_override(tmp_path):
    """The automatic choice (no option set): full search -> the matrix-core sweep k_sweep_q ("sweep" 6), except launches too
    small to pay for fragment prep.  FIC_SWEEP=<n> overrides it process-wide: 5 = the VALU-only sweeps north_star describes
    (k_sweep_d4 where built, else k_sweep_fast), 3 = the exact-covariance matrix-core sweeps.  Same codebooks every time."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import fic_amd; from fic_amd import synth\n"
        "out = []\n"
        "for size, planes, B, n_iso in [(64, 1, 8, 8), (256, 1, 8, 8), (1024, 1, 8, 1), (512, 2, 16, 8)]:\n"
        "    g = np.stack([synth.image_s(size, size, 3 + p) for p in range(planes)])\n"
        "    with fic_amd.Encoder(size, size, B, None, n_iso, planes) as e:\n"
        "        e.set_gray(g); e.encode(); r = e.results(); out.append((e.info()['sweep_kind'], r['qrows'].ravel(), r['iso'].ravel()))\n"
        "np.save(sys.argv[1], np.array([o[0] for o in out]))\n"
        "np.save(sys.argv[2], np.concatenate([o[1] for o in out] + [o[2] for o in out]))\n"
    ) % os.path.dirname(os.path.dirname(GOLDEN))
    res = {}
    for tag, env in (("auto", {}), ("valu", {"FIC_SWEEP": "5"}), ("mfma", {"FIC_SWEEP": "3"})):
        k, q = str(tmp_path / f"k_{tag}.npy"), str(tmp_path / f"q_{tag}.npy")
        e = {kk: v for kk, v in os.environ.items() if kk != "FIC_SWEEP"}
        subprocess.check_call([sys.executable, "-c", code, k, q], env={**e, **env})
        res[tag] = (np.load(k), np.load(q))
    assert res["auto"][0].tolist() == [5, 6, 6, 6]       # 64x64: 64 ranges x 169 blocks stays on the VALU sweep
    assert res["valu"][0].tolist() == [5, 5, 2, 2]       # k_sweep_d4 at B = 8 with 8 isometries, else k_sweep_fast
    assert res["mfma"][0].tolist() == [3, 3, 3, 3]
    assert (res["auto"][1] == res["valu"][1]).all() and (res["auto"][1] == res["mfma"][1]).all()


# ---- n_iso = 8 at B = 4 and B = 16 (k_sweep_mfma<1>, <8>) ------------------------------------------------------

ISO8_OTHER = [(n, B) for n in sorted(IMAGES) for B in (4, 16)
              if not (n in ("lena64", "flat64", "U200", "S200") and B == 16) and not (n in ("lena256", "S256", "U256") and B == 4)]


@pytest.mark.parametrize("name,B", ISO8_OTHER)
def test_mfma_8iso_other_block_sizes_match_oracle(oracle, name, B):
    g = IMAGES[name]
    h, w = g.shape
    Dw = fic_amd.geometry(w, h, B)[2]
    ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, Dw, 8)
    for chunks in (0, 2):
        got = fic_amd.encode_gray(g, B, None, 8, sweep=3, chunks=chunks)
        assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
        assert (got["iso"] == ref["iso"]).all()
        assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
        assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
        assert same_f32(got["err"], ref["err"])


@pytest.mark.parametrize("B,size", [(4, 512), (16, 1024)])
def test_mfma_8iso_other_block_sizes_equal_valu_sweep_at_scale_and_sharded(B, size):
    imgs = np.stack([synth.image_u(size, size, 41 + B), synth.image_s(size, size, 42 + B)])
    with fic_amd.Encoder(size, size, B, None, 8, planes=2) as enc:
        enc.set_gray(imgs)
        enc.set_option("sweep", 2)
        enc.encode()
        valu = enc.results()
        enc.set_option("sweep", 3)
        parts = []
        for b, c in fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3):   # odd tile offsets: misaligned groups at B = 4
            enc.encode(b, c)
            r = enc.results()
            parts.append({k: v[:, b:b + c].copy() for k, v in r.items()})
    _same({k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}, valu)


# ---- "sweep" = 4: the i8-operand kernels at B = 4 / 8 (where "sweep" = 3 now runs the bf16-operand kernels) -----------

@pytest.mark.parametrize("B,n_iso,size", [(4, 8, 256), (8, 8, 512), (4, 1, 256), (8, 1, 512), (16, 8, 512), (16, 1, 512)])
def test_i8_and_bf16_operand_kernels_agree_on_one_context(B, n_iso, size):
    """One context, the sweeps switched back and forth (the fragment stores are rebuilt when the operand type
    changes): VALU == matrix-core/bf16 == matrix-core/i8, bit for bit."""
    imgs = np.stack([synth.image_u(size, size, 41 + B), synth.image_s(size, size, 42 + B)])
    res = {}
    with fic_amd.Encoder(size, size, B, None, n_iso, planes=2) as enc:
        enc.set_gray(imgs)
        for i, sweep in enumerate((2, 3, 4, 3)):
            enc.set_option("sweep", sweep)
            enc.encode()
            res[i] = {k: v.copy() for k, v in enc.results().items()}
            assert enc.info()["sweep_kind"] == sweep
    for i in (1, 2, 3):
        _same(res[0], res[i])


@pytest.mark.parametrize("name", ["lena64", "S128", "U200"])
@pytest.mark.parametrize("n_iso", [1, 8])
def test_i8_operand_kernels_match_oracle(oracle, name, n_iso):
    g = IMAGES[name]
    h, w = g.shape
    for B in (4, 8):
        Dw = fic_amd.geometry(w, h, B)[2]
        ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, Dw, n_iso)
        got = fic_amd.encode_gray(g, B, None, n_iso, sweep=4, chunks=2)
        assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
        assert (got["iso"] == ref["iso"]).all()
        assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
        assert same_f32(got["err"], ref["err"])
