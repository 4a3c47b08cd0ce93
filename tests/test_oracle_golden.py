"""Pins the CPU oracle (oracle/fic_oracle.c) against everything the reference itself holds for
this path:
  K1  unknown.run <-> LenaColored.jpg   full 1024-row RGB codebook, byte for byte
  K2  the five Animation.gif "MSE" labels on LenaGrey.png (grey encode -> quantise -> decode)
plus the .run hashes of SURVEY.md's table (an independent numpy restatement made by the survey)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

K2 = json.load(open(os.path.join(GOLDEN, "k2_animation_gif.json")))["cases"]


def test_k1_unknown_run_byte_identical(oracle, lena_colored):
    ref = open(os.path.join(GOLDEN, "unknown_run.bin"), "rb").read()
    assert hashlib.sha256(ref).hexdigest().startswith("940ad9d6")
    hdr = np.frombuffer(ref[:20], dtype=">i4")
    assert list(hdr) == [1, 256, 256, 8, 2]
    argb = oracle.rgb_to_argb(lena_colored)
    assert not oracle.is_greyscale(argb, 256, 256)
    info = oracle.encode_rgb(argb, 256, 256, 8, 2)
    assert oracle.write_run_rgb(info, 256, 256, 8, 2) == ref


@pytest.mark.parametrize("case", K2, ids=[f"B{c['B']}_wK{c['wK']}" for c in K2])
def test_k2_animation_gif_mse_labels(oracle, lena_grey, case):
    argb = oracle.gray_to_argb(lena_grey)
    assert oracle.is_greyscale(argb, 256, 256)
    e = oracle.encode_gray(argb, 256, 256, case["B"], case["wK"])
    run = oracle.write_run_gray(e["info"], 256, 256, case["B"], case["wK"])
    img, avg, iters = oracle.decode_gray(run)
    # the GUI label is Float.toString(avgError): the shortest decimal that identifies the float
    assert avg == np.float32(case["mse_label"])
    assert float(avg) * 65536.0 == case["ssd"]
    assert iters == 7


SURVEY_TABLE = [  # image, B, wK, sha256 prefix of the .run stream, final SSD, iterations
    ("lena_grey", 16, 16, "8c4fd364a6961c1f", 24537, 7),
    ("lena_grey", 8, 16, "1eb3f3b26ec48dd6", 23906, 7),
    ("lena_grey", 4, 16, "c0c572d4a9e45ace", 48340, 7),
    ("lena_grey", 8, 8, "21036056a93812dc", 34344, 7),
    ("lena_grey", 8, 4, "0af6f55e901c9c15", 23840, 7),
    ("lena_grey", 8, 2, "3889c3adad799f79", 38426, 6),
    ("lena_grey", 8, 61, "a15285e1e53f8d1c", 16463, 7),
    ("lena64", 4, 29, "26d834e11a12cf2c", 1062, 7),
    ("lena64", 4, 2, "be645dad359afccf", 1979, 6),
]


@pytest.mark.parametrize("name,B,wK,sha,ssd,iters", SURVEY_TABLE)
def test_survey_hash_table(oracle, lena_grey, lena64, name, B, wK, sha, ssd, iters):
    g = lena_grey if name == "lena_grey" else lena64
    h, w = g.shape
    e = oracle.encode_gray(oracle.gray_to_argb(g), w, h, B, wK)
    run = oracle.write_run_gray(e["info"], w, h, B, wK)
    assert len(run) == 20 + 12 * (w // B) * (h // B)
    assert hashlib.sha256(run).hexdigest().startswith(sha)
    img, avg, it = oracle.decode_gray(run)
    assert it == iters and float(avg) * w * h == ssd


def test_java_int_cast(oracle):
    L = oracle.lib()
    assert L.fo_java_f2i(float("nan")) == 0
    assert L.fo_java_f2i(3.99) == 3 and L.fo_java_f2i(-3.99) == -3
    assert L.fo_java_f2i(1e20) == 2**31 - 1 and L.fo_java_f2i(-1e20) == -(2**31)


def test_range_subset_equals_full(oracle, lena64):
    argb = oracle.gray_to_argb(lena64)
    full = oracle.encode_gray(argb, 64, 64, 4, 29)
    part = oracle.encode_gray(argb, 64, 64, 4, 29, r0=100, r1=140)
    assert (part["info"][100:140].view(np.uint32) == full["info"][100:140].view(np.uint32)).all()
    assert (part["info"][:100] == 0).all()


def test_iso1_is_reference_and_iso8_never_worse(oracle, lena64):
    argb = oracle.gray_to_argb(lena64)
    e1 = oracle.encode_gray(argb, 64, 64, 4, 29, n_iso=1)
    e8 = oracle.encode_gray(argb, 64, 64, 4, 29, n_iso=8)
    assert (e1["iso"] == 0).all()
    assert (e8["err"] <= e1["err"]).all()
    assert e8["iso"].max() > 0


def test_scale_quirk_uses_height(oracle):
    """FC:993 compares x+1 with image.height: on a wide image the 4th tap becomes 128."""
    w, h = 16, 8
    g = np.full((h, w), 200, np.uint8)
    out = np.zeros((h // 2) * (w // 2), np.int32)
    argb = oracle.gray_to_argb(g)
    import ctypes as C
    oracle.lib().fo_scale_image(argb.ctypes.data_as(C.POINTER(C.c_int32)), w, h, out.ctypes.data_as(C.POINTER(C.c_int32)))
    s = ((out.view(np.uint32) >> 16) & 0xFF).reshape(h // 2, w // 2)
    assert (s[:, :3] == 200).all()            # x+1 < 8
    assert (s[:, 4:] == (600 + 128) // 4).all()  # x+1 >= 8


def test_decode_rows_twin_equals_literal_decoder(oracle, lena_grey):
    """fo_decode_rows (the isometry-aware twin used for n_iso = 8) == fo_decode_gray when iso is absent."""
    argb = oracle.gray_to_argb(lena_grey)
    for B, wK in [(8, 16), (16, 16), (8, 2)]:
        e = oracle.encode_gray(argb, 256, 256, B, wK)
        run = oracle.write_run_gray(e["info"], 256, 256, B, wK)
        a = oracle.decode_gray(run)
        b = oracle.decode_rows(oracle.quantise_gray(e["info"]), None, 256, 256, B, wK)
        c = oracle.decode_rows(oracle.quantise_gray(e["info"]), np.zeros(e["info"].shape[0], np.int32), 256, 256, B, wK)
        for x in (b, c):
            assert (x[0] == a[0]).all() and x[1] == a[1] and x[2] == a[2]
