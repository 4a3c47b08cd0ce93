"""The kernels bench.py times, asserted at the geometries it times them at.

  * headline: 64 x 512x512, B=8, full search, 8 isometries -- the batch bench.py builds (same seeds), through the
    library default (sweep 0) and through sweep 6 explicitly: must be k_sweep_q in ONE pool chunk (the <4, 2, false>
    instantiation), equal to the VALU sweep (sweep 2) on every range block of every plane, and equal to the oracle on a
    sample of range blocks scanned over the full 15 625-block pool;
  * the same batch with 1 isometry (the reference algorithm; k_sweep_q<4, 0, false>);
  * config 5's shape: 24 x 1024x1024, B=8, 8 isometries (k_sweep_q<4, 2, false> over 2001 domain tiles), against the
    VALU sweep on every range block and against two oracle rows."""
import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import same_f32

pytestmark = pytest.mark.gpu


def _batch(seed_key, w, h, planes, rank=0):
    s = synth.SEEDS[seed_key]
    return np.stack([synth.image_u(w, h, s + 3 * (rank * planes + p)) for p in range(planes)])


def _encode(imgs, B, n_iso, sweep):
    P, H, W = imgs.shape
    with fic_amd.Encoder(W, H, B, None, n_iso, P) as enc:
        if sweep:
            enc.set_option("sweep", sweep)
        enc.set_gray(imgs)
        enc.encode()
        r = enc.results()
        return r, enc.info()


def _same_codebooks(a, b):
    for k in ("idx_local", "idx_global", "iso", "qrows"):
        bad = np.argwhere(a[k] != b[k])
        assert bad.size == 0, f"{k}: {len(bad)} mismatches, first at {bad[0]}"
    for k in ("a", "b", "err"):
        assert same_f32(a[k], b[k]), k


def _oracle_rows(oracle, img, B, n_iso, lo, hi, got_plane):
    h, w = img.shape
    Dw = 2 * (w // B) - 3
    ref = oracle.encode_gray(oracle.gray_to_argb(img), w, h, B, Dw, n_iso, lo, hi)
    sl = slice(lo, hi)
    assert (got_plane["idx_local"][sl] == ref["info"][sl, 0].astype(np.int32)).all()
    assert (got_plane["iso"][sl] == ref["iso"][sl]).all()
    assert same_f32(got_plane["a"][sl], ref["info"][sl, 1]) and same_f32(got_plane["b"][sl], ref["info"][sl, 2])
    assert (got_plane["qrows"][sl] == oracle.quantise_gray(ref["info"][sl])).all()
    assert same_f32(got_plane["err"][sl], ref["err"][sl])


@pytest.mark.parametrize("n_iso", [8, 1])
def test_bench_headline_batch_through_the_timed_kernel(oracle, n_iso):
    imgs = _batch("cfg2", 512, 512, 64)
    valu, iv = _encode(imgs, 8, n_iso, 2)
    assert iv["sweep_kind"] == 2
    for sweep in (0, 6):
        got, info = _encode(imgs, 8, n_iso, sweep)
        assert info["sweep_kind"] == 6, f"sweep={sweep}: the library ran kind {info['sweep_kind']}, not k_sweep_q"
        assert info["chunks"] == 1, f"sweep={sweep}: {info['chunks']} pool chunks; bench.py's launch has one (k_sweep_q<4, *, false>)"
        _same_codebooks(got, valu)
        if sweep == 0:
            for p, (lo, hi) in ((0, (2000, 2000 + (16 if n_iso == 8 else 64))), (63, (4090, 4096)), (17, (0, 6))):
                _oracle_rows(oracle, imgs[p], 8, n_iso, lo, hi, {k: v[p] for k, v in got.items()})


def test_config5_shape_through_the_timed_kernel(oracle):
    imgs = _batch("cfg5", 1024, 1024, 24)
    valu, iv = _encode(imgs, 8, 8, 2)
    assert iv["sweep_kind"] == 2
    got, info = _encode(imgs, 8, 8, 0)
    assert info["sweep_kind"] == 6 and info["chunks"] == 1
    _same_codebooks(got, valu)
    _oracle_rows(oracle, imgs[0], 8, 8, 5000, 5001, {k: v[0] for k, v in got.items()})
    _oracle_rows(oracle, imgs[23], 8, 8, 16383, 16384, {k: v[23] for k, v in got.items()})


def test_a_fresh_context_can_encode_at_once_on_a_non_blocking_stream(oracle):
    """Round-3 finding (bench.py's check context with two processes on one GPU): fic_ctx_create zero-fills the pool with
    hipMemset, which only enqueues on the null stream; an encode issued right away on a NON-BLOCKING stream could be overtaken
    by the fill.  Here the null stream is kept busy by a long sweep of another context while a new context is created and
    used immediately on a torch (non-blocking) stream."""
    import torch
    big = synth.image_u(2048, 2048, 3)
    g = synth.image_u(256, 256, synth.SEEDS["cfg3"])
    want = fic_amd.encode_gray(g, 8, None, 8, sweep=2)
    s = torch.cuda.Stream()
    with fic_amd.Encoder(2048, 2048, 4, None, 1) as busy:
        busy.set_option("sweep", 2)                      # VALU sweep: tens of milliseconds
        busy.set_gray(big)
        busy.encode()                                    # warm: allocations, code object
        busy.sync()
        for _ in range(3):
            busy.encode(0, -1, None)                     # the null stream is busy from here on
            with fic_amd.Encoder(256, 256, 8, None, 8) as enc:
                enc.set_gray(torch.from_numpy(g).cuda().view(1, 256, 256))
                enc.set_option("sweep", 2)
                enc.encode(0, -1, s)
                got = {k: v[0] for k, v in enc.results().items()}
            _same_codebooks(got, want)
        busy.sync()
