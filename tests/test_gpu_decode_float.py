"""The decoder's avgError is a Java float accumulation, one add per pixel in range-block order (FractalCompression.java:
385-407).  The GPU decoder sums exact integers and uses them only when that is provably the same number; otherwise -- an
iteration whose sum reaches 2^24, a codebook that never converges (50 iterations, FC:381), a non-integer carry-in of the
static avgError -- it re-accumulates sequentially in Java's order.  These inputs force that path; the reference values
come from the oracle's restatement of the Java loop."""
import ctypes as C

import numpy as np
import pytest

import fic_amd
from fic_amd import capi

pytestmark = pytest.mark.gpu


def _run_gray(w, h, B, wK, rows):
    hdr = b"".join(int(v).to_bytes(4, "big", signed=True) for v in (0, w, h, B, wK))
    return hdr + np.asarray(rows, ">i4").tobytes()


def _decode_debug(run, w, h, avg_in=0.0):
    buf = np.frombuffer(run, np.uint8)
    out = np.zeros(w * h, np.uint8)
    avg = C.c_float(avg_in)
    it, seq = C.c_int(), C.c_int()
    capi.check(capi.lib().fic_debug_decode_gray_run(capi.ptr(buf, C.c_uint8), buf.size, 0, capi.ptr(out, C.c_uint8), out.size,
                                                    C.byref(avg), C.byref(it), C.byref(seq)))
    return out.reshape(h, w), np.float32(avg.value), it.value, seq.value


@pytest.mark.parametrize("size,B", [(512, 8), (256, 4), (512, 16)])
def test_non_converging_codebook_matches_java_float_accumulation(oracle, size, B):
    """a = -1 makes every range block the negative of its domain block plus ~300: the image oscillates, every iteration's
    sum is ~5e8 (far above 2^24) and the loop runs all 50 iterations; the returned avgError is that of iteration 49 (FC:416)."""
    rng = np.random.default_rng(size + B)
    nr = (size // B) ** 2
    rows = np.zeros((nr, 3), np.int64)
    rows[:, 0] = rng.integers(0, 4, nr)                     # window-local index, wK = 2
    rows[:, 1] = -100                                       # a = q / 100 = -1: x -> b - x, eigenvalue -1 on the constant mode
    rows[:, 2] = rng.integers(270, 331, nr)                 # 128 -> ~172 -> ~128 -> ...: squared change ~2000 per pixel, for ever
    run = _run_gray(size, size, B, 2, rows)
    want_img, want_avg, want_it = oracle.decode_gray(run)
    img, avg, it, seq = _decode_debug(run, size, size)
    assert want_it == 50 and it == 50
    assert seq > 0                                           # the sequential path really ran
    assert (img == want_img).all()
    assert np.float32(avg).view(np.uint32) == np.float32(want_avg).view(np.uint32)
    # and through the public entry
    img2, avg2, it2 = fic_amd.decode_gray_run(run)
    assert (img2 == want_img).all() and it2 == 50 and np.float32(avg2).view(np.uint32) == np.float32(want_avg).view(np.uint32)


def test_non_integer_carry_in_takes_the_sequential_sum(oracle):
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "lena_grey_256.npy"))
    e = oracle.encode_gray(oracle.gray_to_argb(g), 256, 256, 8, 4, 1)
    run = oracle.write_run_gray(e["info"], 256, 256, 8, 4)
    for carry in (0.36376953, 1234.567, 7.0):
        want_img, want_avg, want_it = oracle.decode_gray(run, carry)
        img, avg, it, seq = _decode_debug(run, 256, 256, carry)
        assert (img == want_img).all() and it == want_it
        assert np.float32(avg).view(np.uint32) == np.float32(want_avg).view(np.uint32)
        assert seq > 0            # iteration 0 (from the flat grey start) changes every pixel by tens of levels: sum >= 2^24


def test_non_converging_rgb_codebook(oracle):
    rng = np.random.default_rng(5)
    w = h = 256
    B, wK = 8, 2
    nr = (w // B) ** 2
    rows = np.zeros((nr, 5), np.int64)
    rows[:, 0] = rng.integers(0, 4, nr)
    rows[:, 1] = -1000000                                         # a = q / 1e6 = -1
    rows[:, 2] = rng.integers(270, 331, nr) * 100000              # bR = q / 1e5
    rows[:, 3] = rng.integers(270, 331, nr) * 100000
    rows[:, 4] = rng.integers(270, 331, nr)
    hdr = b"".join(int(v).to_bytes(4, "big", signed=True) for v in (1, w, h, B, wK))
    run = hdr + rows.astype(">i4").tobytes()
    want, want_avg, want_it = oracle.decode_rgb(run)
    argb, avg, it, ww, hh = fic_amd.decode_rgb_run(run)
    u = argb.view(np.uint32)
    got = np.stack([(u >> 16) & 0xFF, (u >> 8) & 0xFF, u & 0xFF], -1).reshape(h, w, 3)
    assert it == want_it == 50
    assert (got == want).all()
    assert np.float32(avg).view(np.uint32) == np.float32(want_avg).view(np.uint32)


def _seq_sum(carry, vals):
    s = np.float32(carry)
    for v in vals.astype(np.float32):
        s = np.float32(s + v)
    return s


@pytest.mark.parametrize("case", ["squares", "ties", "powers", "rgb_max", "small", "fractional", "crossing"])
def test_float_accumulation_scan_equals_the_sequential_loop(case):
    """java_float_sum (parallel scan over parity-dependent rounding maps) against `s = fl(s + v)` in order, on inputs
    chosen to hit round-to-even ties, binade crossings inside a block, and a fractional carry-in."""
    rng = np.random.default_rng(11)
    n = 300000
    carry = 0.0
    if case == "squares":
        vals = rng.integers(0, 256, n) ** 2
    elif case == "ties":            # multiples of 2^j: exactly half an ulp once the sum is in the right binade
        vals = (rng.integers(0, 64, n) << rng.integers(0, 12, n)).astype(np.int64)
    elif case == "powers":
        vals = 1 << rng.integers(0, 18, n)
    elif case == "rgb_max":
        vals = np.full(n, 3 * 255 * 255)
    elif case == "small":
        vals = rng.integers(0, 3, 40000)
    elif case == "fractional":
        carry = 0.36376953
        vals = rng.integers(0, 3000, n)
    else:                           # sums hovering around 2^24, 2^25, ...: crossings in the middle of 16384-value blocks
        vals = np.concatenate([np.full(16000, 1000), rng.integers(0, 5, 50000), np.full(20000, 1700), rng.integers(0, 9, 50000)])
    vals = np.ascontiguousarray(vals, np.uint32)
    if case in ("small", "powers"):
        vals = vals[:-3]            # a length that is not a multiple of 4
    out = C.c_float()
    capi.check(capi.lib().fic_debug_float_sum(0, C.c_float(carry), capi.ptr(vals, C.c_uint32), vals.size, C.byref(out)))
    assert np.float32(out.value).view(np.uint32) == _seq_sum(carry, vals).view(np.uint32)
