"""Joint-RGB encode on the GPU (encodeRGB FC:171-219) against the K1-pinned oracle -- and K1 itself:
the reference's own committed encoder output unknown.run, reproduced byte for byte by the GPU."""
import hashlib
import io
import os

import numpy as np
import pytest

import fic_amd
from fic_amd import synth
from conftest import GOLDEN, same_f32

pytestmark = pytest.mark.gpu


def _rgb_synth(w, h, seed, flat=False):
    """Integer-only synthetic colour image: three U planes (S planes when flat=True)."""
    f = synth.image_s if flat else synth.image_u
    return np.stack([f(w, h, seed), f(w, h, seed + 1), f(w, h, seed + 2)], axis=-1)


def _check(oracle, rgb, B, wK):
    h, w = rgb.shape[:2]
    argb = oracle.rgb_to_argb(rgb)
    ref = oracle.encode_rgb(argb, w, h, B, wK)
    got = fic_amd.encode_rgb(argb, w, h, B, wK, want_collage=True)
    assert (got["idx_local"] == ref[:, 0].astype(np.int32)).all()
    for k, col in (("a", 1), ("bR", 2), ("bG", 3), ("bB", 4)):
        assert same_f32(got[k], ref[:, col]), k
    assert (got["qrows"] == oracle.quantise_rgb(ref)).all()
    assert fic_amd.write_run_rgb(got["qrows"], w, h, B, wK) == oracle.write_run_rgb(ref, w, h, B, wK)
    assert (got["collage"] == oracle.collage_rgb(argb, w, h, B, wK, ref)).all()
    return got


def test_k1_unknown_run_reproduced_by_the_gpu(lena_colored, oracle):
    """K1: LenaColored.jpg, B=8, wK=2 (the GUI defaults) -> the reference's committed unknown.run."""
    ref = open(os.path.join(GOLDEN, "unknown_run.bin"), "rb").read()
    argb = oracle.rgb_to_argb(lena_colored)
    got = fic_amd.encode_rgb(argb, 256, 256, 8, 2)
    run = fic_amd.write_run_rgb(got["qrows"], 256, 256, 8, 2)
    assert run == ref
    assert hashlib.sha256(run).hexdigest().startswith("940ad9d6")


def test_k1_through_the_drop_in_entry_point(lena_colored, oracle):
    """FractalCompression.encode(colour RasterImage, out) dispatches to encodeRGB (FC:55-58)."""
    fc = fic_amd.FractalCompression
    fc.blockgroesse, fc.widthKernel = 8, 2
    img = fic_amd.RasterImage(256, 256, oracle.rgb_to_argb(lena_colored))
    out = io.BytesIO()
    collage = fc.encode(img, out)
    assert out.getvalue() == open(os.path.join(GOLDEN, "unknown_run.bin"), "rb").read()
    ref = oracle.encode_rgb(img.argb, 256, 256, 8, 2)
    assert (collage.argb == oracle.collage_rgb(img.argb, 256, 256, 8, 2, ref)).all()
    assert same_f32(fc.imageInfoRGB, ref)


@pytest.mark.parametrize("B,wK", [(8, 2), (8, 4), (8, 16), (4, 2), (4, 16), (16, 2), (16, 8), (8, 61), (16, 29)])
def test_lena_colored_windows_and_full_search(lena_colored, oracle, B, wK):
    _check(oracle, lena_colored, B, wK)


@pytest.mark.parametrize("B,wK,flat", [(8, 2, False), (8, 29, False), (4, 5, True), (8, 29, True), (16, 13, True), (4, 61, False)])
def test_synthetic_colour_images(oracle, B, wK, flat):
    _check(oracle, _rgb_synth(128, 128, 77, flat), B, wK)


@pytest.mark.parametrize("size,B,flat", [(200, 8, False), (200, 8, True), (200, 4, True), (256, 4, False), (256, 8, True)])
def test_full_search_fast_sweep(oracle, size, B, flat):
    """Full search at B = 4 / 8 runs k_sweep_rgb_fast (lane = range block, pool chunks across workgroups): N_r not a
    multiple of 64 (200x200: 625 / 2500 range blocks), several pool chunks, flat blocks and ties on the S planes."""
    Dw = fic_amd.geometry(size, size, B)[2]
    _check(oracle, _rgb_synth(size, size, 31 + B, flat), B, Dw)


def test_non_square_and_extremes(oracle):
    _check(oracle, _rgb_synth(192, 128, 5, True), 8, 5)          # W > H: the FC:940 height quirk
    _check(oracle, _rgb_synth(128, 192, 6), 8, 3)
    sat = np.zeros((64, 64, 3), np.uint8)
    sat[::2, :, 0] = 255
    sat[:, ::2, 1] = 255
    sat[32:, :, 2] = 255                                          # large |greyR*greyD| sums: f32 accumulation order
    _check(oracle, sat, 16, 5)
    _check(oracle, sat, 8, 13)
    _check(oracle, np.full((64, 64, 3), (10, 200, 30), np.uint8), 8, 13)   # constant colour: 0/0 fits


def test_rgb_errors():
    argb = np.zeros(64 * 64, np.int32)
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.encode_rgb(argb, 64, 64, 8, 14)
    assert e.value.code == -2
    with pytest.raises(fic_amd.FicError):
        fic_amd.encode_rgb(argb, 64, 62, 8, 2)


@pytest.mark.parametrize("B,wK", [(8, 2), (8, 16), (4, 4), (16, 8), (8, 61)])
def test_decode_rgb_matches_reference_decoder(lena_colored, oracle, B, wK):
    """decodeRGB (FC:430-508) on the GPU vs the oracle: image, avgError bits, iterations."""
    argb = oracle.rgb_to_argb(lena_colored)
    run = oracle.write_run_rgb(oracle.encode_rgb(argb, 256, 256, B, wK), 256, 256, B, wK)
    want, wavg, wit = oracle.decode_rgb(run)
    got, avg, it, w, h = fic_amd.decode_rgb_run(run)
    u = got.view(np.uint32)
    rgb = np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], axis=-1).astype(np.uint8).reshape(h, w, 3)
    assert (w, h) == (256, 256) and it == wit
    assert (rgb == want).all()
    assert np.float32(avg).view(np.uint32) == np.float32(wavg).view(np.uint32)


def test_k1_stream_decodes_through_the_mirror(oracle):
    """FractalCompression.decode(stream) on the reference's own unknown.run (isRGB = 1, FC:547-553)."""
    run = open(os.path.join(GOLDEN, "unknown_run.bin"), "rb").read()
    fc = fic_amd.FractalCompression
    fc.avgError = np.float32(0.0)
    img = fc.decode(io.BytesIO(run))
    want, wavg, _ = oracle.decode_rgb(run)
    u = img.argb.view(np.uint32)
    rgb = np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], axis=-1).astype(np.uint8).reshape(256, 256, 3)
    assert (rgb == want).all() and fc.getAvgError() == wavg
    assert (img.argb.view(np.uint32) >> 24 == 0xFF).all()


def test_config5_per_channel_encode(lena_colored, oracle):
    """Config 5's "RGB, per-channel encode": three independent grey planes per image."""
    batch = np.stack([lena_colored[:128, :128], lena_colored[128:, 128:]])
    got = fic_amd.encode_rgb_per_channel(batch, 8, None, 8)
    assert got["idx_local"].shape == (2, 3, 256)
    for i in range(2):
        for c in range(3):
            g = np.ascontiguousarray(batch[i, :, :, c])
            ref = oracle.encode_gray(oracle.gray_to_argb(g), 128, 128, 8, 29, 8)
            assert (got["idx_local"][i, c] == ref["info"][:, 0].astype(np.int32)).all()
            assert (got["iso"][i, c] == ref["iso"]).all()
            assert same_f32(got["a"][i, c], ref["info"][:, 1]) and same_f32(got["b"][i, c], ref["info"][:, 2])


@pytest.mark.parametrize("B,wK", [(8, 2), (8, None), (4, 5)])
def test_rgb_context_batched_planes_equal_one_shot(oracle, lena_colored, B, wK):
    """fic_rgb_ctx_*: a batch of colour images in one device-resident context (config-5 style) gives, per image, the bits
    of the one-shot entry; the context decodes its own codebook (decodeRGB) to what the oracle's decoder produces."""
    import torch
    size = 128 if wK is None else 256
    imgs = [lena_colored[:size, :size], _rgb_synth(size, size, 41), _rgb_synth(size, size, 52, flat=True)]
    argbs = [oracle.rgb_to_argb(np.ascontiguousarray(x)) for x in imgs]
    Dw = fic_amd.geometry(size, size, B)[2]
    wk = Dw if wK is None else wK
    with fic_amd.capi.RgbEncoder(size, size, B, wk, planes=3) as enc:
        enc.set_argb(np.stack(argbs))
        enc.encode(with_collage=True)
        r = enc.results()
        dec, avg, it = enc.decode()
        # the same batch from a device-resident input, without the collage
        t = torch.from_numpy(np.stack(argbs)).cuda()
        enc.set_argb(t)
        enc.encode(with_collage=False, stream=torch.cuda.current_stream())
        enc.sync()
        r2 = enc.results()
    for p, argb in enumerate(argbs):
        one = fic_amd.encode_rgb(argb, size, size, B, wk, want_collage=True)
        for k in ("idx_local", "qrows"):
            assert (r[k][p] == one[k]).all() and (r2[k][p] == one[k]).all()
        for k in ("a", "bR", "bG", "bB"):
            assert same_f32(r[k][p], one[k]) and same_f32(r2[k][p], one[k])
        assert (r["collage"][p] == one["collage"]).all()
        run = fic_amd.write_run_rgb(r["qrows"][p], size, size, B, wk)
        want, wavg, wit = oracle.decode_rgb(run)
        u = dec[p].view(np.uint32)
        got = np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], -1).reshape(size, size, 3)
        assert (got == want).all() and it[p] == wit and avg[p].view(np.uint32) == np.float32(wavg).view(np.uint32)
    fic_amd.capi.release_cache()
