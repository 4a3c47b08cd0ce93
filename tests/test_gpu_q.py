"""k_sweep_q (fic_q.hip, "sweep" = 6, the library default for full search): the matrix cores compute an APPROXIMATE
|cov|/sqrt(var) (normalised f16 domain operand) that only decides which pairs are evaluated exactly.  Same bar as every
other sweep -- bit-identical codebooks against the oracle -- plus inputs that force its own rare branches (rule 26 of the
CDNA guide: a rare data-dependent branch needs an input that takes it):
  * low-contrast range blocks: the first domain tile of a chunk cannot seed theta out of order (L < 0.26 n) and is evaluated whole;
  * letterboxed images: whole domain tiles flat before anything was evaluated (theta still "none");
  * near-ties and exact duplicates: pairs within the error bound E_r of the running best must all be evaluated;
  * every chunk count: each chunk start re-seeds theta; candidate 0 must always be evaluated (all-ties fallback winner)."""
import os

import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import GOLDEN, same_f32

pytestmark = pytest.mark.gpu


def _letterbox(size, seed, bar):
    g = synth.image_u(size, size, seed).copy()
    g[:bar] = 16
    g[-bar:] = 16
    return g


def _low_contrast(size, seed):
    """+-1/+-2 texture on slow ramps: ||r - rM|| of a few units, so L < 0.26 n for every pair of most ranges."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]
    return (100 + (x // 37) + (y // 53) + rng.integers(-2, 3, (size, size))).astype(np.uint8)


def _smooth_low(size, seed, amp):
    """Slow waves of a few grey levels + +-1 noise: smooth blocks as natural images have them, ||r - rM|| of 3..25 -- the best
    pair's L lies between 0.26 rem (from where a pair may seed / publish theta in any index order) and 0.26 n (round 2's bound)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size]
    return np.rint(120 + amp * np.sin(x / 9.0) + amp * np.cos(y / 7.0) + 0.5 * amp * np.sin((x + 2 * y) / 23.0)
                   + rng.integers(-1, 2, (size, size))).astype(np.uint8)


def _duplicates(size, seed):
    """A 16x16 tile repeated everywhere: every domain block occurs dozens of times (exact ties, lowest index must win)."""
    rng = np.random.default_rng(seed)
    return np.tile(rng.integers(0, 256, (16, 16)).astype(np.uint8), (size // 16, size // 16))


def _one_noisy_block(size, seed):
    g = np.full((size, size), 90, np.uint8)
    g[size // 2:size // 2 + 16, size // 2:size // 2 + 16] = synth.image_u(16, 16, seed)
    return g


IMAGES = {
    "lena64": np.load(os.path.join(GOLDEN, "lena64.npy")),
    "lena256": np.load(os.path.join(GOLDEN, "lena_grey_256.npy")),
    "U128": synth.image_u(128, 128, synth.SEEDS["cfg2"]),
    "S128": synth.image_s(128, 128, synth.SEEDS["cfg2"]),
    "S256": synth.image_s(256, 256, synth.SEEDS["cfg3"]),
    "U200": synth.image_u(200, 200, 11),     # partial range group and partial domain tile
    "flat64": np.full((64, 64), 77, np.uint8),
    "letterbox": _letterbox(192, 5, 40),
    "lowcontrast": _low_contrast(160, 6),
    "smooth3": _smooth_low(160, 9, 3.0),
    "smooth8": _smooth_low(128, 10, 8.0),
    "duplicates": _duplicates(128, 7),
    "onenoisy": _one_noisy_block(128, 8),
    "ramp": ((np.arange(160)[None, :] * 3 + np.arange(160)[:, None] * 5) % 256).astype(np.uint8),
}

_CACHE = {}


def _ref(oracle, name, B, n_iso):
    key = (name, B, n_iso)
    if key not in _CACHE:
        g = IMAGES[name]
        h, w = g.shape
        Dw = fic_amd.geometry(w, h, B)[2]
        _CACHE[key] = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, Dw, n_iso)
    return _CACHE[key]


def _check(oracle, got, ref):
    want = ref["info"][:, 0].astype(np.int32)
    bad = np.nonzero(got["idx_local"] != want)[0]
    assert bad.size == 0, f"{bad.size} index mismatches, first at range {bad[0]}: got {got['idx_local'][bad[0]]} want {want[bad[0]]}"
    assert (got["iso"] == ref["iso"]).all()
    assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
    assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()
    assert same_f32(got["err"], ref["err"])


CASES = [(n, B, k) for n in sorted(IMAGES) for B in (4, 8, 16) for k in (1, 8)
         if not (B == 16 and n in ("lena64", "flat64", "U200")) and not (B == 4 and n in ("lena256", "S256", "letterbox"))]


@pytest.mark.parametrize("name,B,n_iso", CASES)
def test_q_sweep_matches_oracle(oracle, name, B, n_iso):
    g = IMAGES[name]
    ref = _ref(oracle, name, B, n_iso)
    for chunks in (0, 1, 3):
        got = fic_amd.encode_gray(g, B, None, n_iso, sweep=6, chunks=chunks)
        _check(oracle, got, ref)


@pytest.mark.parametrize("name,B,n_iso", [("lowcontrast", 8, 8), ("letterbox", 8, 1), ("duplicates", 8, 8), ("S128", 4, 8), ("U128", 16, 1),
                                          ("smooth3", 8, 8), ("smooth8", 8, 1), ("smooth3", 4, 1), ("lena256", 8, 8)])
def test_q_sweep_many_chunk_starts(oracle, name, B, n_iso):
    """Up to one chunk per domain tile: every tile is a chunk's first tile (out-of-order seeding everywhere)."""
    g = IMAGES[name]
    ref = _ref(oracle, name, B, n_iso)
    for chunks in (7, 50, 10000):
        got = fic_amd.encode_gray(g, B, None, n_iso, sweep=6, chunks=chunks)
        _check(oracle, got, ref)


def test_q_is_the_default_batched_and_sharded(oracle):
    imgs = [IMAGES["lena256"], IMAGES["S256"], synth.image_u(256, 256, 77)]
    for n_iso in (1, 8):
        with fic_amd.Encoder(256, 256, 8, None, n_iso, planes=3) as enc:
            enc.set_gray(np.stack(imgs))
            enc.encode()
            assert enc.info()["sweep_kind"] == 6
            whole = {k: v.copy() for k, v in enc.results().items()}
            spans = fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3)
            parts = []
            for b, c in spans:                               # three logical shards, results gathered per span
                enc.encode(b, c)
                r = enc.results()
                parts.append({k: v[:, b:b + c].copy() for k, v in r.items()})
        for k in ("idx_local", "iso", "qrows"):
            assert (np.concatenate([p[k] for p in parts], axis=1) == whole[k]).all()
        for p, g in enumerate(imgs):
            Dw = fic_amd.geometry(256, 256, 8)[2]
            ref = oracle.encode_gray(oracle.gray_to_argb(g), 256, 256, 8, Dw, n_iso)
            _check(oracle, {k: v[p] for k, v in whole.items()}, ref)


def test_q_records_are_the_packed_results():
    import torch
    g = IMAGES["S256"]
    with fic_amd.Encoder(256, 256, 8, None, 8) as enc:
        enc.set_gray(g)
        enc.encode()
        r = enc.results()
        rec = enc.records_device().cpu().numpy()
    back = fic_amd.unpack_records(rec)
    for k in ("idx_local", "iso", "qrows"):
        assert (back[k] == r[k]).all()
    assert same_f32(back["a"], r["a"]) and same_f32(back["b"], r["b"])


def _encode_shape(g, B, shape, chunks=0):
    h, w = g.shape
    with fic_amd.Encoder(w, h, B, None, 1) as enc:
        enc.set_option("sweep", 6)
        enc.set_option("q_shape", shape)
        if chunks:
            enc.set_option("chunks", chunks)
        enc.set_gray(g)
        enc.encode()
        return {k: v[0] for k, v in enc.results().items()}, enc.last_kernel()


@pytest.mark.parametrize("name,B", [(n, B) for n in sorted(IMAGES) for B in (8, 16) if not (B == 16 and n in ("lena64", "flat64", "U200"))])
def test_q16_sweep_matches_oracle(oracle, name, B):
    """k_sweep_q16: the 1-isometry sweep on v_mfma_f32_16x16x32_f16 (the library takes it for large pools only; option
    "q_shape" = 1 forces it here) -- every image of this file, chunk counts from one to one per domain tile, against the
    oracle; and the 32x32x16 kernel on the same inputs."""
    g = IMAGES[name]
    ref = _ref(oracle, name, B, 1)
    for chunks in (0, 1, 3, 50, 10000):
        got, kname = _encode_shape(g, B, 1, chunks)
        assert kname.startswith((f"k_sweep_q16<{B * B // 16}, ", f"k_sweep_q16s<{B * B // 16}>")), kname     # (k_sweep_q16s: short pool chunks)
        _check(oracle, got, ref)
    got, kname = _encode_shape(g, B, 2)
    assert kname.startswith((f"k_sweep_q<{B * B // 16}, 0, ", f"k_sweep_qs<{B * B // 16}, 0>")), kname
    _check(oracle, got, ref)


def test_q16_is_chosen_by_pool_size_and_batches_and_shards():
    """Default choice: small pools keep 32x32x16; a 4096x4096 pool at B = 8 takes k_sweep_q16 (tests/test_gpu_fullsize.py compares
    that whole codebook with the VALU sweep).  Batched planes and range shards through the forced 16x16x32 kernel."""
    imgs = np.stack([IMAGES["lena256"], IMAGES["S256"], synth.image_u(256, 256, 77)])
    with fic_amd.Encoder(256, 256, 8, None, 1, planes=3) as enc:
        enc.set_gray(imgs)
        enc.encode()
        assert enc.last_kernel().startswith(("k_sweep_q<4, 0, ", "k_sweep_qs<4, 0>"))
        whole = {k: v.copy() for k, v in enc.results().items()}
        enc.set_option("q_shape", 1)
        enc.encode()
        assert enc.last_kernel().startswith(("k_sweep_q16<4, ", "k_sweep_q16s<4>"))
        forced = {k: v.copy() for k, v in enc.results().items()}
        parts = []
        for b, c in fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, 3):
            enc.encode(b, c)
            parts.append({k: v[:, b:b + c].copy() for k, v in enc.results().items()})
    for k in ("idx_local", "idx_global", "iso", "qrows"):
        assert (forced[k] == whole[k]).all() and (np.concatenate([p[k] for p in parts], axis=1) == whole[k]).all()
    for k in ("a", "b", "err"):
        assert same_f32(forced[k], whole[k]) and same_f32(np.concatenate([p[k] for p in parts], axis=1), whole[k])
    with fic_amd.Encoder(4096, 4096, 8, None, 1) as enc:
        enc.set_gray(np.zeros((4096, 4096), np.uint8))
        enc.encode()
        assert enc.last_kernel().startswith("k_sweep_q16<4, ")


def test_small_launches_run_as_two_kernels(oracle):
    """A small launch of the default sweep (fic_q_prep_fused: B = 8, at most 1024 prep workgroups) is k_prep_q8 + the sweep, whose
    last workgroup per (plane, range-column group) finalises that group's range blocks (q_finalize_tail, fic_q.hip) -- no k_scale /
    k_pool_q / k_range_q8 / k_finalize launches.  Checked: the library says so (last_kernel suffix); results equal the VALU
    sweep's (whose finaliser is k_finalize) for 8 and 1 isometries, one and many pool chunks, both MFMA shapes, repeated encodes
    on one context (the group counters must be back at zero), range shards that begin and end inside a column group, and two
    planes; a 64-plane launch is not small and keeps the separate kernels."""
    FUSED = " after k_prep_q8, finalising"
    g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
    for n_iso in (8, 1):
        want = fic_amd.encode_gray(g, 8, None, n_iso, sweep=2)
        with fic_amd.Encoder(512, 512, 8, None, n_iso) as enc:
            enc.set_gray(g)
            seen = set()
            for chunks, shape in ((0, 0), (0, 0), (1, 0), (7, 0), (500, 0), (0, 1), (3, 1), (0, 0)):
                if n_iso == 8 and shape:
                    continue
                enc.set_option("chunks", chunks)
                enc.set_option("q_shape", shape)
                enc.encode()
                assert enc.last_kernel().endswith(FUSED), enc.last_kernel()
                seen.add(enc.last_kernel()[:-len(FUSED)])
                got = {k: v[0] for k, v in enc.results().items()}
                for k in ("idx_local", "idx_global", "iso", "qrows"):
                    assert (got[k] == want[k]).all(), (n_iso, chunks, shape, k)
                for k in ("a", "b", "err"):
                    assert same_f32(got[k], want[k]), (n_iso, chunks, shape, k)
            # one pool chunk, long chunks (theta_g read in flagged tiles), short chunks (k_sweep_qs: theta_g refreshed in the fast path)
            mode = 2 if n_iso == 8 else 0
            assert {f"k_sweep_q<4, {mode}, false>", f"k_sweep_q<4, {mode}, true>", f"k_sweep_qs<4, {mode}>"} <= seen, seen
            if n_iso == 1:
                assert {"k_sweep_q16s<4>", "k_sweep_q16<4, true>"} <= seen, seen
            enc.set_option("chunks", 0)
            enc.set_option("q_shape", 0)
            # shards: spans of whole range tiles, and one ragged span inside a tile
            tile = enc.ranges_per_tile
            for b, c in list(fic_amd.shard_spans(enc.n_ranges, tile, 5)) + [(tile + 3, 2 * tile + 11)]:
                enc.encode(b, c)
                assert enc.last_kernel().endswith(FUSED)
                got = {k: v[0, b:b + c] for k, v in enc.results().items()}
                for k in ("idx_local", "idx_global", "iso", "qrows"):
                    assert (got[k] == want[k][b:b + c]).all(), (n_iso, b, c, k)
                for k in ("a", "b", "err"):
                    assert same_f32(got[k], want[k][b:b + c]), (n_iso, b, c, k)
    imgs = np.stack([IMAGES["lena256"], IMAGES["S256"]])
    with fic_amd.Encoder(256, 256, 8, None, 8, planes=2) as enc:
        enc.set_gray(imgs)
        enc.encode()
        assert enc.last_kernel().endswith(FUSED)
        got = enc.results()
    for p, name in enumerate(("lena256", "S256")):
        _check(oracle, {k: v[p] for k, v in got.items()}, _ref(oracle, name, 8, 8))
    with fic_amd.Encoder(256, 256, 8, None, 8, planes=64) as enc:
        enc.set_gray(np.stack([IMAGES["lena256"]] * 64))
        enc.encode()
        assert enc.last_kernel() == "k_sweep_q<4, 2, false>" or enc.last_kernel() == "k_sweep_q<4, 2, true>", enc.last_kernel()


def test_q16_long_chunks_on_a_batch_of_512_pools():
    """Round-3 incident (profiles/r03W_*): a build of k_sweep_q16<4, true> (several LONG pool chunks, 16x16x32 MFMA) lost the winners
    of ~20 % of the range blocks of every wave's third column tile -- only on pools of this size (489 domain tiles in 2-3 chunks),
    which no test ran through that kernel: the small images of this file give short chunks, the 4096x4096 ones thousands of tiles.
    40 images through both MFMA shapes with 2 and 3 chunks, separate k_finalize (not a small launch), against the VALU sweep."""
    P = 40
    imgs = np.stack([synth.image_u(512, 512, synth.SEEDS["cfg2"] + 3 * p) for p in range(P)])
    with fic_amd.Encoder(512, 512, 8, None, 1, P) as enc:
        enc.set_gray(imgs)
        enc.set_option("sweep", 2)
        enc.encode()
        want = {k: v.copy() for k, v in enc.results().items()}
        enc.set_option("sweep", 6)
        for chunks, shape, name in ((3, 1, "k_sweep_q16<4, true>"), (2, 1, "k_sweep_q16<4, true>"), (3, 2, "k_sweep_q<4, 0, true>"), (1, 1, "k_sweep_q16<4, false>")):
            enc.set_option("chunks", chunks)
            enc.set_option("q_shape", shape)
            enc.encode()
            assert enc.last_kernel() == name, enc.last_kernel()
            got = enc.results()
            for k in ("idx_local", "idx_global", "iso", "qrows"):
                bad = np.argwhere(got[k] != want[k])
                assert bad.size == 0, f"{name} chunks={chunks}: {k}: {len(bad)} mismatches, by column tile in wave {np.bincount((bad[:, 1] % 128) // 32, minlength=4)}"
            for k in ("a", "b", "err"):
                assert same_f32(got[k], want[k]), (name, chunks, k)


@pytest.mark.parametrize("B,n_iso,size,planes", [(8, 8, 512, 12), (16, 1, 512, 12), (16, 8, 512, 12), (4, 1, 256, 12), (4, 8, 256, 12), (8, 1, 1024, 2)])
def test_long_chunks_on_medium_pools_every_instantiation(B, n_iso, size, planes):
    """The same coverage for the other instantiations: pools of a few hundred to a few thousand domain tiles in 2 / 3 / 5 chunks
    (long chunks: theta_g read in the flagged tiles, prefix seed from tile 0), several planes, both MFMA shapes where there are two,
    against the VALU sweep."""
    imgs = np.stack([synth.image_u(size, size, 4000 + 7 * p + B + n_iso) for p in range(planes)])
    with fic_amd.Encoder(size, size, B, None, n_iso, planes) as enc:
        enc.set_gray(imgs)
        enc.set_option("sweep", 2)
        enc.encode()
        want = {k: v.copy() for k, v in enc.results().items()}
        enc.set_option("sweep", 6)
        for shape in ((1, 2) if (n_iso == 1 and B >= 8) else (0,)):
            enc.set_option("q_shape", shape)
            for chunks in (2, 3, 5):
                enc.set_option("chunks", chunks)
                enc.encode()
                got = enc.results()
                for k in ("idx_local", "idx_global", "iso", "qrows"):
                    bad = np.argwhere(got[k] != want[k])
                    assert bad.size == 0, f"{enc.last_kernel()} chunks={chunks}: {k}: {len(bad)} mismatches, first {bad[0]}"
                for k in ("a", "b", "err"):
                    assert same_f32(got[k], want[k]), (enc.last_kernel(), chunks, k)
