"""Parity tests proper: the HIP path (through the C ABI of libfic_hip.so) against the CPU oracle
on the same inputs.  Bar: bit-exact -- integer indices, quantised rows, and the float32 bits
of the unquantised a/b (every NaN counted equal: Java has a single NaN value).

Inputs: the reference's own images (tests/golden: Lena64, LenaGrey -> K2 known answers run
through the GPU encoder), and the synthetic U / S images of SURVEY.md 8(d) -- S is the
adversarial one (flat blocks, rem == 0 ranges, exact ties, NaN fits)."""
import ctypes as C
import hashlib
import io
import json
import os

import numpy as np
import pytest

import fic_amd
from fic_amd import capi, synth
from conftest import GOLDEN, same_f32

pytestmark = pytest.mark.gpu


_ORACLE_CACHE = {}


def _oracle_encode(oracle, g, B, wK, n_iso, r0=0, r1=None):
    h, w = g.shape
    key = (hashlib.sha256(g.tobytes()).hexdigest(), w, h, B, wK, n_iso, r0, r1)
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, wK, n_iso, r0, r1)
    return _ORACLE_CACHE[key]


def _assert_same(oracle, got, ref, lo=0, hi=None):
    hi = ref["info"].shape[0] if hi is None else hi
    sl = slice(lo, hi)
    want_idx = ref["info"][sl, 0].astype(np.int32)
    bad = np.nonzero(got["idx_local"][sl] != want_idx)[0]
    assert bad.size == 0, f"{bad.size} idx mismatches, first at range {lo + bad[0]}: got {got['idx_local'][lo + bad[0]]} want {want_idx[bad[0]]}"
    assert (got["iso"][sl] == ref["iso"][sl]).all()
    assert same_f32(got["a"][sl], ref["info"][sl, 1])
    assert same_f32(got["b"][sl], ref["info"][sl, 2])
    assert (got["qrows"][sl] == oracle.quantise_gray(ref["info"][sl])).all()
    assert same_f32(got["err"][sl], ref["err"][sl])


def _images():
    return {
        "lena64": np.load(os.path.join(GOLDEN, "lena64.npy")),
        "lena256": np.load(os.path.join(GOLDEN, "lena_grey_256.npy")),
        "U128": synth.image_u(128, 128, synth.SEEDS["cfg2"]),
        "S128": synth.image_s(128, 128, synth.SEEDS["cfg2"]),
        "S256": synth.image_s(256, 256, synth.SEEDS["cfg3"]),
        "U256": synth.image_u(256, 256, synth.SEEDS["cfg3"]),
        "flat64": np.full((64, 64), 77, np.uint8),
        "wide": synth.image_s(192, 128, 7),          # W > H: exercises the FC:993 height quirk
        "tall": synth.image_u(128, 192, 9),
    }


IMAGES = _images()


def test_device_present():
    assert capi.lib().fic_device_count() >= 1


def test_f64_sqrt_is_correctly_rounded_for_every_variance():
    """Domainblock.variance is an integer in [0, 256*255^2] < 2^24: check sqrt on ALL of them
    against the host's correctly rounded sqrt (Java Math.sqrt is correctly rounded too)."""
    n = 1 << 24
    out = np.zeros(n + 1, np.float64)
    capi.check(capi.lib().fic_debug_sqrt_f64(0, 0, n + 1, out.ctypes.data_as(C.POINTER(C.c_double))))
    want = np.sqrt(np.arange(n + 1, dtype=np.float64))
    assert (out.view(np.uint64) == want.view(np.uint64)).all()


@pytest.mark.parametrize("name", ["lena64", "S128", "U128", "wide", "tall", "lena256"])
@pytest.mark.parametrize("B", [4, 8, 16])
def test_pool_build_matches_createCodebuch(oracle, name, B):
    g = IMAGES[name]
    h, w = g.shape
    if (w // B) * 2 - 3 < 2 or (h // B) * 2 - 3 < 2:
        pytest.skip("too small for this B")
    pix, mean, var = oracle.pool(oracle.gray_to_argb(g), w, h, B)
    with fic_amd.Encoder(w, h, B, wK=1) as enc:
        enc.set_gray(g)
        enc.encode()
        p = enc.debug_pool()
    assert (p["pix"][0].astype(np.int32) == pix).all()
    assert (p["sum"][0] // (B * B) == mean).all()
    assert (p["var"][0].astype(np.float32) == var).all()


FULL_CASES = [  # image, B, n_iso
    ("lena64", 4, 1), ("lena64", 4, 8), ("lena64", 8, 1), ("lena64", 8, 8),
    ("U128", 4, 1), ("U128", 8, 1), ("U128", 8, 8), ("U128", 16, 1), ("U128", 16, 8),
    ("S128", 4, 1), ("S128", 4, 8), ("S128", 8, 1), ("S128", 8, 8), ("S128", 16, 1), ("S128", 16, 8),
    ("flat64", 4, 1), ("flat64", 8, 8),
    ("lena256", 8, 1), ("lena256", 8, 8), ("lena256", 16, 1), ("lena256", 16, 8), ("lena256", 4, 1),
    ("S256", 8, 1), ("S256", 8, 8), ("U256", 8, 1), ("U256", 16, 8),
]


@pytest.mark.parametrize("name,B,n_iso", FULL_CASES)
@pytest.mark.parametrize("sweep", [0, 2, 1], ids=["default", "fast", "generic"])
def test_full_search_matches_oracle(oracle, name, B, n_iso, sweep):
    """Config 1 (Lena64, B=4, full) and friends: full search == widthKernel = Dw (FC:89-96)."""
    g = IMAGES[name]
    h, w = g.shape
    Rw, Rh, Dw, Dh = fic_amd.geometry(w, h, B)
    ref = _oracle_encode(oracle, g, B, Dw, n_iso)
    got = fic_amd.encode_gray(g, B, None, n_iso, sweep=sweep)
    _assert_same(oracle, got, ref)


@pytest.mark.parametrize("chunks", [1, 2, 3, 7, 50])
@pytest.mark.parametrize("name,B,n_iso", [("S128", 8, 8), ("U128", 4, 1), ("lena256", 16, 8), ("S256", 8, 1)])
def test_chunking_of_the_pool_does_not_change_the_result(oracle, name, B, n_iso, chunks):
    g = IMAGES[name]
    h, w = g.shape
    Dw = fic_amd.geometry(w, h, B)[2]
    ref = _oracle_encode(oracle, g, B, Dw, n_iso)
    got = fic_amd.encode_gray(g, B, None, n_iso, sweep=2, chunks=chunks)
    _assert_same(oracle, got, ref)


WINDOW_CASES = [("lena256", 8, 2, 1), ("lena256", 8, 4, 1), ("lena256", 8, 8, 1), ("lena256", 8, 16, 1),
                ("lena256", 16, 16, 1), ("lena256", 4, 16, 1), ("lena64", 4, 2, 1), ("lena64", 4, 2, 8),
                ("S128", 8, 4, 8), ("wide", 8, 5, 1), ("tall", 8, 3, 8), ("wide", 16, 2, 1), ("S256", 4, 16, 1)]


@pytest.mark.parametrize("name,B,wK,n_iso", WINDOW_CASES)
def test_window_search_matches_oracle(oracle, name, B, wK, n_iso):
    """The GUI's real settings: local windows of 2/4/8/16 (CTL:142), incl. non-square images."""
    g = IMAGES[name]
    ref = _oracle_encode(oracle, g, B, wK, n_iso)
    got = fic_amd.encode_gray(g, B, wK, n_iso)
    _assert_same(oracle, got, ref)


K2 = json.load(open(os.path.join(GOLDEN, "k2_animation_gif.json")))["cases"]


@pytest.mark.parametrize("case", K2, ids=[f"B{c['B']}_wK{c['wK']}" for c in K2])
def test_k2_gui_labels_through_the_gpu_encoder(oracle, case):
    """Reference known-answer: encode LenaGrey on the GPU, decode with the reference decoder
    restatement, and land on the exact 'MSE' label the reference GUI shows in Animation.gif."""
    g = IMAGES["lena256"]
    got = fic_amd.encode_gray(g, case["B"], case["wK"])
    run = fic_amd.write_run_gray(got["qrows"], 256, 256, case["B"], case["wK"])
    img, avg, iters = oracle.decode_gray(run)
    assert avg == np.float32(case["mse_label"])


SHA = [("lena256", 8, 61, "a15285e1e53f8d1c"), ("lena64", 4, 29, "26d834e11a12cf2c"), ("lena256", 8, 2, "3889c3adad799f79")]


@pytest.mark.parametrize("name,B,wK,sha", SHA)
def test_run_stream_hashes(name, B, wK, sha):
    g = IMAGES[name]
    got = fic_amd.encode_gray(g, B, wK)
    run = fic_amd.write_run_gray(got["qrows"], g.shape[1], g.shape[0], B, wK)
    assert hashlib.sha256(run).hexdigest().startswith(sha)


def test_drop_in_entry_point_and_collage(oracle):
    """FractalCompression.encode(RasterImage, out) -> .run bytes + collage RasterImage (FC:54, 160-161)."""
    g = IMAGES["lena256"]
    fc = fic_amd.FractalCompression
    for B, wK in [(8, 2), (8, 61), (16, 16)]:
        fc.blockgroesse, fc.widthKernel, fc.n_iso = B, wK, 1
        out = io.BytesIO()
        collage = fc.encode(fic_amd.RasterImage.from_gray(g), out)
        ref = _oracle_encode(oracle, g, B, wK, 1)
        assert out.getvalue() == oracle.write_run_gray(ref["info"], 256, 256, B, wK)
        want = oracle.collage_gray(oracle.gray_to_argb(g), 256, 256, B, wK, ref["info"])
        assert (collage.argb == want).all()
        assert same_f32(fc.imageInfo, ref["info"])
    fc.blockgroesse, fc.widthKernel = 8, 2


def test_mirror_public_methods_beside_encode_and_decode(oracle):
    """The other public members of FractalCompression on the path: getBestGeneratedCollage (FC:269-300, what
    encodeGrayScale returns), decodeGreyScale / decodeRGB (FC:356, 430: the stream positioned after the isRGB int)."""
    g = IMAGES["lena64"]
    fc = fic_amd.FractalCompression
    fc.blockgroesse, fc.widthKernel, fc.n_iso = 4, 5, 1
    img = fic_amd.RasterImage.from_gray(g)
    out = io.BytesIO()
    collage = fc.encode(img, out)
    assert (fc.getBestGeneratedCollage(img).argb == collage.argb).all()
    run = out.getvalue()
    fc.avgError = np.float32(0.0)
    a = fc.decode(io.BytesIO(run))
    avg_a = fc.getAvgError()
    fc.avgError = np.float32(0.0)
    b = fc.decodeGreyScale(io.BytesIO(run[4:]))
    assert (a.argb == b.argb).all() and fc.getAvgError() == avg_a
    with pytest.raises(fic_amd.FicError):
        fc.getBestGeneratedCollage(fic_amd.RasterImage(128, 128))
    fc.blockgroesse, fc.widthKernel = 8, 2
    fc.avgError = np.float32(0.0)


def test_one_shot_c_entry_points(oracle):
    g = IMAGES["lena64"]
    ref = _oracle_encode(oracle, g, 4, 29, 1)
    n = 256
    for use_argb in (False, True):
        idx, a, b = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        iso, q = np.zeros(n, np.int32), np.zeros((n, 3), np.int32)
        if use_argb:
            argb = oracle.gray_to_argb(g)
            rc = capi.lib().fic_encode_gray_argb(capi.ptr(argb, C.c_int32), 64, 64, 4, 29, 1, 0, capi.ptr(idx, C.c_int32),
                                                 capi.ptr(a, C.c_float), capi.ptr(b, C.c_float), capi.ptr(iso, C.c_int32),
                                                 capi.ptr(q, C.c_int32))
        else:
            gg = np.ascontiguousarray(g)
            rc = capi.lib().fic_encode_gray_u8(capi.ptr(gg, C.c_uint8), 64, 64, 4, 29, 1, 0, capi.ptr(idx, C.c_int32),
                                               capi.ptr(a, C.c_float), capi.ptr(b, C.c_float), capi.ptr(iso, C.c_int32),
                                               capi.ptr(q, C.c_int32))
        capi.check(rc)
        assert (idx == ref["info"][:, 0].astype(np.int32)).all()
        assert same_f32(a, ref["info"][:, 1]) and same_f32(b, ref["info"][:, 2])
        assert (q == oracle.quantise_gray(ref["info"])).all()


def test_error_behaviour_matches_reference_failures():
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.encode_gray(IMAGES["lena64"], 4, 30)      # wK > Dw: negative index at FC:145
    assert e.value.code == -2
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.encode_gray(IMAGES["lena64"][:, :62], 4)
    assert e.value.code == -1
    with pytest.raises(fic_amd.FicError):
        fic_amd.encode_gray(IMAGES["wide"], 8, None)      # full search needs a square grid


def test_batched_planes_equal_single_images(oracle):
    """Config 5 shape: independent grey planes in one context."""
    imgs = [synth.image_u(128, 128, 100 + i) if i % 2 else synth.image_s(128, 128, 100 + i) for i in range(5)]
    stack = np.stack(imgs)
    for n_iso in (1, 8):
        with fic_amd.Encoder(128, 128, 8, None, n_iso, planes=5) as enc:
            enc.set_gray(stack)
            enc.encode()
            r = enc.results()
        for i, g in enumerate(imgs):
            ref = _oracle_encode(oracle, g, 8, 29, n_iso)
            _assert_same(oracle, {k: v[i] for k, v in r.items()}, ref)


@pytest.mark.parametrize("shards", [2, 3, 8])
@pytest.mark.parametrize("B,n_iso", [(8, 8), (4, 1)])
def test_logical_shards_on_one_gpu_equal_unsharded(shards, B, n_iso):
    """results are independent of the number of shards by construction."""
    g = IMAGES["U256"]
    whole = fic_amd.encode_gray(g, B, None, n_iso)
    with fic_amd.Encoder(256, 256, B, None, n_iso) as enc:
        enc.set_gray(g)
        spans = fic_amd.shard_spans(enc.n_ranges, enc.ranges_per_tile, shards)
        parts = []
        for b, c in spans:
            enc.encode(b, c)
            r = enc.results()
            parts.append({k: v[0][b:b + c].copy() for k, v in r.items()})
    for k in ("idx_local", "iso", "qrows"):
        assert (np.concatenate([p[k] for p in parts]) == whole[k]).all()
    assert same_f32(np.concatenate([p["a"] for p in parts]), whole["a"])
    assert same_f32(np.concatenate([p["b"] for p in parts]), whole["b"])


def test_device_resident_input_and_result_tensors(oracle):
    import torch
    g = IMAGES["S128"]
    ref = _oracle_encode(oracle, g, 8, 29, 8)
    t = torch.from_numpy(g.copy()).cuda()
    with fic_amd.Encoder(128, 128, 8, None, 8) as enc:
        enc.set_gray(t.view(1, 128, 128))
        enc.set_option("time_sweep", 1)
        enc.encode(stream=torch.cuda.current_stream())
        enc.sync()
        ms, n = enc.sweep_time()
        assert n == 1 and ms > 0
        d = enc.results_device()
        assert d["qrows"].is_cuda and tuple(d["qrows"].shape) == (1, 256, 3)
        rec = fic_amd.pack_records(d, 0, 256).cpu().numpy()
        back = fic_amd.unpack_records(rec)
    assert (back["idx_local"][0] == ref["info"][:, 0].astype(np.int32)).all()
    assert same_f32(back["a"][0], ref["info"][:, 1])
    assert (back["iso"][0] == ref["iso"]).all()


def test_cfg2_size_fast_vs_generic_and_oracle_sample(oracle):
    """Config 2 (512x512, B=8, full search): both GPU kernels agree on every range, and a
    sample of ranges is checked against the oracle over the full 15625-block pool."""
    g = synth.image_u(512, 512, synth.SEEDS["cfg2"])
    for n_iso in (1, 8):
        fast = fic_amd.encode_gray(g, 8, None, n_iso, sweep=2)
        gen = fic_amd.encode_gray(g, 8, None, n_iso, sweep=1)
        for k in ("idx_local", "iso", "qrows"):
            assert (fast[k] == gen[k]).all()
        assert same_f32(fast["a"], gen["a"]) and same_f32(fast["b"], gen["b"])
        lo, hi = 2000, 2000 + (96 if n_iso == 1 else 24)
        ref = _oracle_encode(oracle, g, 8, 125, n_iso, lo, hi)
        _assert_same(oracle, fast, ref, lo, hi)


def test_iso8_never_worse_than_iso1_and_decodes(oracle):
    g = IMAGES["lena256"]
    e1 = fic_amd.encode_gray(g, 8, None, 1)
    e8 = fic_amd.encode_gray(g, 8, None, 8)
    assert (e8["err"] <= e1["err"]).all()
    run = fic_amd.write_run_gray(e1["qrows"], 256, 256, 8, 61)
    img, avg, iters = oracle.decode_gray(run)
    assert abs(oracle.psnr(img, g) - 24.823) < 1e-3     # decoded PSNR of SURVEY table row 7


def test_cpp_host_mirror_of_the_java_entry_point(oracle, tmp_path):
    """include/fic_host.hpp: bvk_ss19::FractalCompression::encode(RasterImage, ostream) / decode(istream) driven
    like RLEAppController.openDecodedImage (CTL:172-188) -- .run bytes, collage and decoded image equal the oracle's."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "cpp", "host_mirror_test")
    if not os.path.exists(exe):
        pytest.fail("tests/cpp/host_mirror_test missing: run __graft_entry__.build()")
    g = IMAGES["lena256"]
    raw = tmp_path / "g.raw"
    raw.write_bytes(g.tobytes())
    for B, wK in [(8, 2), (8, 61), (4, 16)]:
        run, col = tmp_path / f"o{B}_{wK}.run", tmp_path / f"c{B}_{wK}.raw"
        subprocess.check_call([exe, "encode", str(raw), "256", "256", str(B), str(wK), str(run), str(col)])
        ref = _oracle_encode(oracle, g, B, wK, 1)
        assert run.read_bytes() == oracle.write_run_gray(ref["info"], 256, 256, B, wK)
        want = oracle.collage_gray(oracle.gray_to_argb(g), 256, 256, B, wK, ref["info"])
        assert (np.frombuffer(col.read_bytes(), np.int32) == want).all()
        dec = tmp_path / "d.raw"
        out = subprocess.check_output([exe, "decode", str(run), str(dec)], text=True).split()
        wimg, wavg, _ = oracle.decode_gray(run.read_bytes())
        got = (np.frombuffer(dec.read_bytes(), np.int32).view(np.uint32) >> 16) & 0xFF
        assert (got.reshape(256, 256) == wimg).all() and np.float32(out[2]) == wavg
    # colour input dispatches to encodeRGB: the reference's own unknown.run comes out, and decodes back
    c = np.load(os.path.join(GOLDEN, "lena_colored_256.npy"))
    argb = oracle.rgb_to_argb(c)
    araw, run, col, dec = tmp_path / "a.raw", tmp_path / "k1.run", tmp_path / "k1c.raw", tmp_path / "k1d.raw"
    araw.write_bytes(argb.tobytes())
    subprocess.check_call([exe, "encode_argb", str(araw), "256", "256", "8", "2", str(run), str(col)])
    assert run.read_bytes() == open(os.path.join(GOLDEN, "unknown_run.bin"), "rb").read()
    subprocess.check_call([exe, "decode", str(run), str(dec)], stdout=subprocess.DEVNULL)
    wrgb, _, _ = oracle.decode_rgb(run.read_bytes())
    u = np.frombuffer(dec.read_bytes(), np.int32).view(np.uint32)
    assert (np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], -1).reshape(256, 256, 3) == wrgb).all()
    # bad geometry -> exception (exit code 1), like the reference's unchecked exceptions
    rc = subprocess.call([exe, "encode", str(raw), "256", "256", "8", "200", str(tmp_path / "x.run"), str(tmp_path / "x.raw")],
                         stderr=subprocess.DEVNULL)
    assert rc == 1


DECODE_CASES = [("lena256", 8, 16), ("lena256", 16, 16), ("lena256", 4, 16), ("lena256", 8, 8), ("lena256", 8, 4),
                ("lena256", 8, 2), ("lena256", 8, 61), ("lena64", 4, 29), ("lena64", 4, 2), ("S128", 8, 29),
                ("U128", 4, 8), ("wide", 8, 5), ("tall", 16, 3), ("flat64", 8, 13)]


@pytest.mark.parametrize("name,B,wK", DECODE_CASES)
def test_gpu_decoder_matches_reference_decoder(oracle, name, B, wK):
    """FractalCompression.decode (FC:547-553, 356-421) on the GPU: image, avgError bits and
    iteration count equal the oracle's restatement of the Java loop."""
    g = IMAGES[name]
    h, w = g.shape
    ref = _oracle_encode(oracle, g, B, wK, 1)
    run = oracle.write_run_gray(ref["info"], w, h, B, wK)
    want_img, want_avg, want_it = oracle.decode_gray(run)
    img, avg, it = fic_amd.decode_gray_run(run)
    assert it == want_it
    assert (img == want_img).all()
    assert np.float32(avg).view(np.uint32) == np.float32(want_avg).view(np.uint32)


@pytest.mark.parametrize("case", K2, ids=[f"B{c['B']}_wK{c['wK']}" for c in K2])
def test_k2_gui_labels_entirely_on_the_gpu(case):
    """Encode AND decode on the GPU through the drop-in mirror, landing on the reference GUI's label."""
    fc = fic_amd.FractalCompression
    fc.blockgroesse, fc.widthKernel, fc.n_iso = case["B"], case["wK"], 1
    fc.avgError = np.float32(0.0)
    out = io.BytesIO()
    fc.encode(fic_amd.RasterImage.from_gray(IMAGES["lena256"]), out)
    img = fc.decode(io.BytesIO(out.getvalue()))
    assert fc.getAvgError() == np.float32(case["mse_label"])
    assert img.width == 256 and img.height == 256
    fc.blockgroesse, fc.widthKernel = 8, 2


def test_decoder_static_avg_error_carry_over(oracle):
    """FractalCompression.avgError is never reset (FC:20,407): a second decode starts from the first one's value."""
    g = IMAGES["lena256"]
    ref = _oracle_encode(oracle, g, 8, 16, 1)
    run = oracle.write_run_gray(ref["info"], 256, 256, 8, 16)
    _, a1, _ = oracle.decode_gray(run)
    _, a2, i2 = oracle.decode_gray(run, float(a1))
    _, g1, _ = fic_amd.decode_gray_run(run)
    _, g2, j2 = fic_amd.decode_gray_run(run, 0, float(g1))
    assert g1 == a1 and g2 == a2 and i2 == j2


def test_decoder_rejects_what_the_reference_cannot_read():
    with pytest.raises(fic_amd.FicError):
        fic_amd.decode_gray_run(b"\x00" * 10)
    hdr = b"".join(int(v).to_bytes(4, "big", signed=True) for v in (0, 64, 64, 4, 29))
    with pytest.raises(fic_amd.FicError):          # rows missing: EOFException in the reference
        fic_amd.decode_gray_run(hdr + b"\x00" * 100)
    rows = np.zeros((256, 3), ">i4")
    rows[:, 0] = 29 * 29 + 5                       # index outside the pool: AIOOBE at FC:394
    with pytest.raises(fic_amd.FicError):
        fic_amd.decode_gray_run(hdr + rows.tobytes())
    rgb = b"".join(int(v).to_bytes(4, "big", signed=True) for v in (1, 64, 64, 4, 29))
    with pytest.raises(fic_amd.FicError):
        fic_amd.decode_gray_run(rgb + b"\x00" * 5120)


@pytest.mark.parametrize("n_iso", [1, 8])
def test_context_decode_with_isometries_and_batches(oracle, n_iso):
    imgs = [IMAGES["lena256"], synth.image_s(256, 256, 5), synth.image_u(256, 256, 6)]
    with fic_amd.Encoder(256, 256, 8, None, n_iso, planes=3) as enc:
        enc.set_gray(np.stack(imgs))
        enc.encode()
        r = enc.results()
        dec, avg, it = enc.decode()
    for p, g in enumerate(imgs):
        want, wavg, wit = oracle.decode_rows(r["qrows"][p], r["iso"][p] if n_iso > 1 else None, 256, 256, 8, 61)
        assert (dec[p] == want).all() and it[p] == wit
        assert avg[p].view(np.uint32) == np.float32(wavg).view(np.uint32)
    if n_iso == 1:
        assert abs(oracle.psnr(dec[0], imgs[0]) - 24.823) < 1e-3


def test_one_shot_context_cache_reuse(oracle):
    """The one-shot entry keeps working sets per geometry: interleaved geometries and changing images
    must give the same results as fresh contexts."""
    from fic_amd import capi as _c
    _c.release_cache()
    imgs = [IMAGES["lena256"], IMAGES["S256"], IMAGES["U256"]]
    for rep in range(2):
        for g in imgs:
            for B, wK, n_iso in [(8, 2, 1), (8, None, 8), (16, 16, 1), (4, None, 1), (8, 4, 8), (16, None, 8)]:
                got = _c.encode_gray_oneshot(g, B, wK, n_iso)
                Dw = fic_amd.geometry(256, 256, B)[2]
                ref = _oracle_encode(oracle, g, B, Dw if wK is None else wK, n_iso)
                assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
                assert (got["iso"] == ref["iso"]).all()
                assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
    _c.release_cache()


TINY = [  # w, h, B, wK, n_iso : minimum geometries (Rw or Rh = 2 -> Dw or Dh = 1; FC:534,539 branches when Rh == 2)
    (16, 16, 8, 1, 1), (16, 16, 8, 1, 8), (8, 8, 4, 1, 1), (32, 32, 16, 1, 8),
    (12, 12, 4, 1, 1), (12, 12, 4, 3, 1), (12, 12, 4, 3, 8), (24, 24, 8, 2, 1),
    (32, 16, 8, 1, 1), (16, 32, 8, 1, 8), (48, 16, 8, 1, 1), (20, 8, 4, 1, 1), (64, 32, 16, 1, 1),
]


@pytest.mark.parametrize("w,h,B,wK,n_iso", TINY)
def test_minimum_geometries(oracle, w, h, B, wK, n_iso):
    g = synth.image_s(w, h, 1000 + w + h) if (w + h) % 3 else synth.image_u(w, h, 1000 + w + h)
    ref = _oracle_encode(oracle, g, B, wK, n_iso)
    Rw, Rh, Dw, Dh = fic_amd.geometry(w, h, B)
    sweeps = [0, 1] + ([2] if wK == Dw == Dh else [])
    if wK == Dw == Dh and (n_iso == 1 or B == 8) and capi.has_xcheck():
        sweeps.append(3)
    if wK == Dw == Dh:
        sweeps.append(6)
    for sweep in sweeps:
        got = fic_amd.encode_gray(g, B, wK, n_iso, sweep=sweep)
        _assert_same(oracle, got, ref)
    run = fic_amd.write_run_gray(got["qrows"], w, h, B, wK)
    if n_iso == 1:
        want = oracle.decode_gray(run)
        img, avg, it = fic_amd.decode_gray_run(run)
        assert (img == want[0]).all() and it == want[2] and np.float32(avg).view(np.uint32) == np.float32(want[1]).view(np.uint32)


def test_seeded_fuzz_geometries_windows_sweeps(oracle):
    """60 random (W, H, B, wK, n_iso, image kind) cases, every applicable sweep kernel, against the oracle."""
    rng = np.random.default_rng(int(os.environ.get("FIC_FUZZ_SEED", "20261004")))
    done = 0
    while done < int(os.environ.get("FIC_FUZZ_CASES", "60")):
        B = int(rng.choice([4, 8, 16]))
        Rw, Rh = int(rng.integers(2, 160 // B + 1)), int(rng.integers(2, 160 // B + 1))
        if rng.random() < 0.5:
            Rh = Rw                                     # square grids: the only ones with a full search (fast / matrix-core sweeps)
        w, h = Rw * B, Rh * B
        Dw, Dh = 2 * Rw - 3, 2 * Rh - 3
        square_full = (Dw == Dh) and rng.random() < 0.5
        wK = Dw if square_full else int(rng.integers(1, min(Dw, Dh, 17) + 1))
        n_iso = int(rng.choice([1, 8]))
        kind = rng.choice(["U", "S", "flat", "ramp", "smooth", "lowc"])
        seed = int(rng.integers(1, 1 << 30))
        if kind == "U":
            g = synth.image_u(w, h, seed)
        elif kind == "S":
            g = synth.image_s(w, h, seed)
        elif kind == "flat":
            g = np.full((h, w), seed % 256, np.uint8)
        elif kind == "smooth":                          # slow waves of a few grey levels + noise: smooth blocks of natural images
            yy, xx = np.mgrid[0:h, 0:w]
            amp = 1.0 + (seed % 12)
            g = np.rint(128 + amp * np.sin(xx / (5.0 + seed % 7)) + amp * np.cos(yy / (4.0 + seed % 5))
                        + np.random.default_rng(seed).integers(-1, 2, (h, w))).astype(np.uint8)
        elif kind == "lowc":                            # a few grey levels of texture on steps
            yy, xx = np.mgrid[0:h, 0:w]
            g = (60 + (seed % 130) + xx // 29 + yy // 41 + np.random.default_rng(seed).integers(0, 2 + seed % 4, (h, w))).astype(np.uint8)
        else:
            g = ((np.arange(w)[None, :] * 3 + np.arange(h)[:, None] * 5 + seed) % 256).astype(np.uint8)
        ref = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, wK, n_iso)
        sweeps = [1]
        if wK == Dw == Dh:
            sweeps += [2, 6]                            # VALU, k_sweep_q
            if capi.has_xcheck():
                sweeps += [3, 4]                        # round 1's matrix-core sweeps (bf16 / i8 by block size; i8)
            if B == 8 and n_iso == 8:
                sweeps.append(5)                        # VALU with algebraic isometries (k_sweep_d4)
            if B >= 8 and n_iso == 1:
                sweeps.append(16)                       # k_sweep_q16: "sweep" = 6 with the 16x16x32 MFMA shape forced
        for sweep in sweeps:
            got = fic_amd.encode_gray(g, B, wK, n_iso, sweep=6 if sweep == 16 else sweep, chunks=int(rng.choice([0, 1, 2, 3, 7, 40, 10000])) if sweep >= 2 else 0,
                                      q_shape=1 if sweep == 16 else 0)
            try:
                _assert_same(oracle, got, ref)
            except AssertionError as e:
                raise AssertionError(f"case w={w} h={h} B={B} wK={wK} n_iso={n_iso} kind={kind} seed={seed} sweep={sweep}: {e}")
        done += 1


def test_concurrent_one_shot_calls_from_threads(oracle):
    """Several host threads encode different images/geometries at once through the one-shot C entry point."""
    from concurrent.futures import ThreadPoolExecutor
    from fic_amd import capi as _c
    jobs = [("lena256", 8, 2, 1), ("S256", 8, None, 8), ("U256", 16, None, 1), ("lena256", 4, 16, 1),
            ("S256", 16, 8, 8), ("U256", 8, None, 1)] * 3

    def work(job):
        name, B, wK, n_iso = job
        return job, _c.encode_gray_oneshot(IMAGES[name], B, wK, n_iso)

    with ThreadPoolExecutor(6) as ex:
        results = list(ex.map(work, jobs))
    for (name, B, wK, n_iso), got in results:
        Dw = fic_amd.geometry(256, 256, B)[2]
        ref = _oracle_encode(oracle, IMAGES[name], B, Dw if wK is None else wK, n_iso)
        assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all()
        assert (got["iso"] == ref["iso"]).all()
        assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
    _c.release_cache()


def _near_tie_image(w, h, seed, density):
    """A 16x16 base tile repeated over the image + sparse +-1 perturbations: thousands of almost identical
    domain blocks (near-ties of |cov|/sqrt(var) at the 1e-3..1e-6 level) and many exactly identical ones."""
    rng = np.random.default_rng(seed)
    tile = rng.integers(40, 216, (16, 16))
    img = np.tile(tile, (h // 16 + 1, w // 16 + 1))[:h, :w].astype(np.int32)
    mask = rng.random((h, w)) < density
    img = img + mask * rng.choice([-1, 1], (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("B,n_iso", [(8, 1), (8, 8), (4, 1), (16, 1), (16, 8), (4, 8)])
@pytest.mark.parametrize("density", [0.0, 0.002, 0.02])
def test_near_tie_images(oracle, B, n_iso, density):
    g = _near_tie_image(128, 128, 7 + B + n_iso, density)
    Dw = fic_amd.geometry(128, 128, B)[2]
    ref = _oracle_encode(oracle, g, B, Dw, n_iso)
    sweeps = [2, 1, 6] + ([3] if (n_iso == 1 or B == 8) else [])
    for sweep in sweeps:
        for chunks in ((0, 3) if sweep >= 2 else (0,)):
            got = fic_amd.encode_gray(g, B, None, n_iso, sweep=sweep, chunks=chunks)
            _assert_same(oracle, got, ref)


def test_sharded_encoder_wrapper_single_rank(oracle):
    """ShardedEncoder without an initialised process group == one rank owning every range block."""
    import torch
    g = IMAGES["S256"]
    ref = _oracle_encode(oracle, g, 8, 61, 8)
    enc = fic_amd.ShardedEncoder(256, 256, 8, None, 8, planes=1, device=0)
    try:
        assert enc.world == 1 and enc.spans == [(0, 1024)]
        enc.set_gray(torch.from_numpy(g.copy()).cuda().view(1, 256, 256))
        enc.encode_local(torch.cuda.current_stream())
        got = enc.gather()
    finally:
        enc.close()
    assert (got["idx_local"][0] == ref["info"][:, 0].astype(np.int32)).all()
    assert (got["iso"][0] == ref["iso"]).all()
    assert same_f32(got["a"][0], ref["info"][:, 1]) and same_f32(got["b"][0], ref["info"][:, 2])
    assert (got["qrows"][0] == oracle.quantise_gray(ref["info"])).all()
