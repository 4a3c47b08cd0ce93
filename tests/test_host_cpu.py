"""CPU-side checks of the product's host layer: the C-ABI library loads and exports every
symbol include/fic.h declares, validates geometry like the reference would throw, writes the
.run stream byte-exactly, and refuses to compute without a GPU (no fallback)."""
import ctypes as C
import io
import subprocess

import numpy as np
import pytest

import fic_amd
from fic_amd import capi


def test_library_exports_every_declared_symbol():
    names = capi.declared_symbols()
    assert len(names) >= 20
    L = C.CDLL(capi.SO_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/fic.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", capi.SO_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(names) <= exported
    # nothing but the C ABI leaks out of the shared object
    assert all(s.startswith("fic_") for s in exported), exported - set(names)


def test_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", capi.SO_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", capi.SO_PATH], capture_output=True, text=True).stdout
    assert "fo_" not in syms


def test_geometry_matches_reference_formulas(oracle):
    for w, h, B in [(64, 64, 4), (256, 256, 8), (256, 256, 16), (512, 512, 8), (2048, 2048, 4), (4096, 4096, 8),
                    (1024, 1024, 8), (384, 256, 8)]:
        assert fic_amd.geometry(w, h, B) == oracle.geometry(w, h, B)
    assert fic_amd.geometry(4096, 4096, 8) == (512, 512, 1021, 1021)


@pytest.mark.parametrize("w,h,B", [(510, 512, 8), (512, 511, 8), (12, 12, 8), (8, 8, 8), (64, 64, 2), (64, 64, 12),
                                   (0, 64, 8), (66, 66, 4 * 5)])
def test_geometry_rejects_what_the_reference_cannot_encode(w, h, B):
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.geometry(w, h, B)
    assert e.value.code == -1


def test_write_run_gray_bytes(oracle):
    rng = np.random.default_rng(1)
    info = np.stack([rng.integers(0, 800, 50).astype(np.float32), rng.uniform(-1, 1, 50).astype(np.float32),
                     rng.uniform(-300, 300, 50).astype(np.float32)], axis=1)
    info[3, 1] = np.nan
    info[3, 2] = np.nan
    q = oracle.quantise_gray(info)
    assert (q[3] == [int(info[3, 0]), 0, 0]).all()
    assert capi.write_run_gray(q, 64, 64, 4, 29) == oracle.write_run_gray(info, 64, 64, 4, 29)
    with pytest.raises(fic_amd.FicError):
        out = np.zeros(10, np.uint8)
        capi.check(int(capi.lib().fic_write_run_gray(capi.ptr(q, C.c_int32), 50, 64, 64, 4, 29,
                                                     capi.ptr(out, C.c_uint8), out.size)))


def test_is_greyscale_mirror(lena_grey, lena_colored, oracle):
    g = fic_amd.RasterImage.from_gray(lena_grey)
    assert fic_amd.FractalCompression.isGreyScale(g)
    c = fic_amd.RasterImage(256, 256, oracle.rgb_to_argb(lena_colored))
    assert not fic_amd.FractalCompression.isGreyScale(c)
    assert (g.argb == oracle.gray_to_argb(lena_grey)).all()


def test_no_cpu_fallback_without_device(lena64):
    """In the build container there is no GPU: every compute entry must fail loudly."""
    if capi.lib().fic_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.encode_gray(lena64, 4)
    assert "no HIP device" in str(e.value)
    fc = fic_amd.FractalCompression
    fc.blockgroesse, fc.widthKernel = 4, 2
    with pytest.raises(fic_amd.FicError):
        fc.encode(fic_amd.RasterImage.from_gray(lena64), io.BytesIO())
    n = 256
    idx, a, b = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    rc = capi.lib().fic_encode_gray_u8(capi.ptr(np.ascontiguousarray(lena64), C.c_uint8), 64, 64, 4, 29, 1, 0,
                                       capi.ptr(idx, C.c_int32), capi.ptr(a, C.c_float), capi.ptr(b, C.c_float),
                                       None, None)
    assert rc == -4


def test_rgb_dispatch_is_loud(lena_colored, oracle):
    """Colour input dispatches to encodeRGB (FC:55-58), which is GPU-backed: no device -> loud failure."""
    if capi.lib().fic_device_count() > 0:
        pytest.skip("a HIP device is present")
    c = fic_amd.RasterImage(256, 256, oracle.rgb_to_argb(lena_colored))
    with pytest.raises(fic_amd.FicError) as e:
        fic_amd.FractalCompression.encode(c, io.BytesIO())
    assert e.value.code == -4
    with pytest.raises(fic_amd.FicError):
        fic_amd.decode_gray_run(b"\x00" * 20 + b"\x00" * 12)


def test_write_run_rgb_bytes(oracle):
    rng = np.random.default_rng(2)
    info = np.stack([rng.integers(0, 4, 40).astype(np.float32)] + [rng.uniform(-1, 1, 40).astype(np.float32)] +
                    [rng.uniform(-300, 300, 40).astype(np.float32) for _ in range(3)], axis=1)
    info[7, 1:] = np.nan
    q = oracle.quantise_rgb(info)
    assert capi.write_run_rgb(q, 256, 256, 8, 2) == oracle.write_run_rgb(info, 256, 256, 8, 2)


def test_shard_spans_cover_and_align():
    for nr, tile, world in [(4096, 64, 8), (262144, 64, 8), (262144, 128, 8), (400, 128, 3), (256, 256, 4), (1024, 64, 1)]:
        spans = fic_amd.shard_spans(nr, tile, world)
        assert len(spans) == world
        pos = 0
        for b, c in spans:
            assert b == pos or c == 0
            assert b % tile == 0 or c == 0
            pos = b + c if c else pos
        assert pos == nr
    assert fic_amd.shard_planes(192, 8) == [(24 * r, 24) for r in range(8)]


def test_synthetic_images_are_pinned():
    """Known answers so that any other-language generator can be checked against the same bytes."""
    u = fic_amd.synth.image_u(8, 2, 1)
    assert u.tolist() == [[145, 151, 29, 110, 99, 189, 99, 158], [174, 8, 80, 148, 196, 106, 135, 93]]
    s = fic_amd.synth.image_s(128, 64, fic_amd.synth.SEEDS["cfg2"])
    assert s.shape == (64, 128) and s.dtype == np.uint8
    assert (s[:32, :32] == 0).all() and s[0, 32] in (64, 65, 66, 67) and (s[32:, 32:64] == 128).all()
    import hashlib
    assert hashlib.sha256(fic_amd.synth.image_u(512, 512, fic_amd.synth.SEEDS["cfg2"]).tobytes()).hexdigest()[:16] == \
        hashlib.sha256(fic_amd.synth.image_u(512, 512, 0xF1C0002).tobytes()).hexdigest()[:16]


def test_pack_unpack_records_roundtrip():
    rng = np.random.default_rng(0)
    P, N = 2, 100
    res = {"idx_local": rng.integers(0, 1000, (P, N)).astype(np.int32),
           "a": rng.uniform(-1, 1, (P, N)).astype(np.float32), "b": rng.uniform(-300, 300, (P, N)).astype(np.float32),
           "iso": rng.integers(0, 8, (P, N)).astype(np.int32), "qrows": rng.integers(-300, 300, (P, N, 3)).astype(np.int32)}
    res["qrows"][..., 0] = res["idx_local"]
    res["a"][0, 5] = np.nan
    rec = fic_amd.pack_records(res, 10, 50)
    assert rec.shape == (P, 50, 6)
    back = fic_amd.unpack_records(rec)
    for k in ("idx_local", "iso", "qrows"):
        assert (back[k] == res[k][:, 10:60]).all()
    assert (back["a"].view(np.uint32) == res["a"][:, 10:60].view(np.uint32)).all()
    assert (back["b"].view(np.uint32) == res["b"][:, 10:60].view(np.uint32)).all()


@pytest.mark.parametrize("kind", ["U", "S"])
@pytest.mark.parametrize("w,h,seed", [(64, 64, 0xF1C0002), (200, 96, 11), (512, 512, 0xF1C0004), (96, 160, 0xF1C0005 + 3 * 7 + 2)])
def test_synthetic_inputs_are_the_same_bytes_in_c_and_python(tmp_path, kind, w, h, seed):
    """the benchmark inputs must be generated identically in C/C++ and Python (no RNG library).
    include/fic_synth.h (through the C++ driver) against fractal-image-compression_amd/synth.py."""
    import os
    drv = os.path.join(os.path.dirname(__file__), "cpp", "host_mirror_test")
    if not os.path.exists(drv):
        pytest.skip("C++ driver not built (run __graft_entry__.build())")
    out = tmp_path / "img.raw"
    subprocess.check_call([drv, "synth", kind, str(w), str(h), hex(seed), str(out)])
    got = np.fromfile(out, np.uint8).reshape(h, w)
    assert (got == fic_amd.synth.image(kind, w, h, seed)).all()


def test_mirror_generate_kernel_and_grey_image(oracle):
    """FractalCompression.generateKernel (FC:84-100) and generateGrayImage (FC:1142-1148) of the Python mirror are host
    logic; the oracle's restatement of the same Java lines agrees on random inputs."""
    import ctypes as C
    L = oracle.lib()
    fc = fic_amd.FractalCompression
    rng = np.random.default_rng(5)
    keep = fc.widthKernel
    try:
        for _ in range(500):
            Dw, Dh = int(rng.integers(1, 70)), int(rng.integers(1, 70))
            wK = int(rng.integers(1, min(Dw, Dh) + 1))
            idx = int(rng.integers(0, Dw * Dh))
            dy, dx = C.c_int(), C.c_int()
            L.fo_generate_kernel(Dw, Dh, idx, wK, C.byref(dy), C.byref(dx))
            fc.widthKernel = wK
            assert fc.generateKernel(Dw, Dh, idx) == [dy.value, dx.value]
    finally:
        fc.widthKernel = keep
    img = fc.generateGrayImage(7, 3)
    assert (img.width, img.height) == (7, 3) and (img.argb.view(np.uint32) == 0xFF808080).all()
    import os
    drv = os.path.join(os.path.dirname(__file__), "cpp", "host_mirror_test")       # the C++ mirror's generateKernel
    if os.path.exists(drv):
        for Dw, Dh, idx, wK in [(29, 29, 0, 5), (61, 61, 1830, 16), (13, 29, 200, 4), (125, 125, 15624, 125)]:
            dy, dx = C.c_int(), C.c_int()
            L.fo_generate_kernel(Dw, Dh, idx, wK, C.byref(dy), C.byref(dx))
            out = subprocess.run([drv, "kernel", str(Dw), str(Dh), str(idx), str(wK)], capture_output=True, text=True).stdout.split()
            assert [int(v) for v in out] == [dy.value, dx.value]


def test_d4_tables_are_verified_and_current(tmp_path):
    """tools/gen_d4_tables.py derives the group-Fourier tables of k_sweep_d4 from the kernels' isometry definition, checks the
    closed form against brute force on random blocks, and must reproduce the committed csrc/fic_d4_tables.h byte for byte."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_d4_tables", os.path.join(root, "tools", "gen_d4_tables.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    t = gen.order_accumulators(gen.build(8))
    packed, counts, U, V = gen.pack(t)
    umax, vmax = gen.verify(t, packed, counts, U, V, trials=100)
    assert len(packed) == 96 and umax < 32768 and vmax < 32768              # 48 dot2 pairs, i16 operands
    # the isometry definition the tables are built from is the oracle's (and the kernels') iso_source
    for k in range(8):
        for x, y in [(0, 0), (3, 1), (7, 2), (5, 5)]:
            assert gen.iso_source(k, 8, x, y) == oracle_iso_source(k, 8, x, y)
    out = tmp_path / "fic_d4_tables.h"
    gen.emit([(t, packed, counts, U, V, umax, vmax)], str(out))
    committed = os.path.join(root, "fractal-image-compression_amd", "csrc", "fic_d4_tables.h")
    assert out.read_text() == open(committed).read()


def oracle_iso_source(k, B, x, y):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import fic_oracle
    return fic_oracle.lib().fo_iso_source(k, B, x, y)


def test_jni_shim_parses_against_a_stub_jni_h():
    """No JDK in the image: the JNI shim cannot be built, but gcc can parse and type-check it against a declaration-only
    stand-in for jni.h (tests/jni_stub/jni.h: the JNI-spec signatures of the JNIEnv entries the shim uses) and the real
    include/fic.h -- so every fic_* call in it matches the C ABI.  Also: the shim must never hold a JNI critical region."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "fractal-image-compression_amd", "jni", "fic_jni.c")
    r = subprocess.run(["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "tests", "jni_stub"),
                        "-I" + os.path.join(root, "include"), src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    code = open(src).read().split("*/", 1)[1]
    assert "PrimitiveArrayCritical" not in code
    java = open(os.path.join(root, "fractal-image-compression_amd", "java", "bvk_ss19", "FicNative.java")).read()
    for name in ("deviceCount", "encodeGray", "encodeGrayMulti", "encodeRgb", "decode"):
        assert f"native " in java and f" {name}(" in java and f"Java_bvk_1ss19_FicNative_{name}(" in code


def test_device_side_synthetic_generator_equals_the_numpy_one():
    """synth.images_u_torch (what bench.py fills HBM with) produces the integers of synth.image_u (SURVEY 8d's U)."""
    import torch
    from fic_amd import synth
    seeds = [synth.SEEDS["cfg2"], synth.SEEDS["cfg5"] + 3 * 17 + 2, 0xFFFFFFFFFFFFFFF0, 7]
    got = synth.images_u_torch(96, 64, seeds, "cpu").numpy()
    for i, s in enumerate(seeds):
        assert (got[i] == synth.image_u(96, 64, s)).all()
    assert got.dtype == np.uint8 and got.shape == (4, 64, 96)
