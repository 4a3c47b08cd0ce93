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


def _seq_sum_fast(carry, vals):
    """The same sequential float32 loop through numpy's accumulate (one rounding per add, in order)."""
    v = np.concatenate([[np.float32(carry)], vals.astype(np.float32)]).astype(np.float32)
    return np.add.accumulate(v, dtype=np.float32)[-1]


@pytest.mark.parametrize("case", ["wrap32", "image16M", "stagnation", "ragged", "huge_carry", "one_segment_crossing"])
def test_multi_workgroup_float_accumulation(case):
    """Round 3: the sum runs on every CU (one map per 65 536-value segment and binade, then an ordered walk); only segments in
    which the sum changes binade take the sequential-order path.  wrap32: segment totals above 2^32 at ulp 1 (the
    tree used to wrap); image16M: as many squares as a 4096x4096 iteration; stagnation: addends below half an ulp leave the sum
    where it is, far behind the exact prefix; ragged: a length that is no multiple of the segment; huge_carry: the static
    avgError (FC:20) arriving large enough that nothing moves it."""
    rng = np.random.default_rng(23)
    carry = 0.0
    if case == "wrap32":
        vals = 65600 + rng.integers(0, 40, 3 * 65536 + 777)
    elif case == "image16M":
        vals = rng.integers(0, 256, 4096 * 4096) ** 2
    elif case == "stagnation":
        vals = np.concatenate([np.full(70000, 190000), np.full(2000000, 20), rng.integers(0, 70, 500000), np.full(300000, 33)])
    elif case == "ragged":
        vals = rng.integers(0, 195076, 5 * 65536 + 12345)
    elif case == "huge_carry":
        carry = 3.0e13
        vals = rng.integers(0, 195076, 200000)
    else:
        vals = np.concatenate([np.full(16000, 1000), rng.integers(0, 5, 30000), np.full(10000, 1700)])
    vals = np.ascontiguousarray(vals, np.uint32)
    out = C.c_float()
    capi.check(capi.lib().fic_debug_float_sum(0, C.c_float(carry), capi.ptr(vals, C.c_uint32), vals.size, C.byref(out)))
    want = _seq_sum_fast(carry, vals)
    assert np.float32(out.value).view(np.uint32) == np.float32(want).view(np.uint32), (out.value, want)
    nseg = (vals.size + 65535) // 65536
    fb = capi.lib().fic_debug_float_sum_fallbacks()
    # only the 4096-value sub-segments in which the sum changes binade take the sequential-order path: at most ~20 of them
    assert fb <= 24, f"{fb} sub-segments (of {16 * nseg}) took the sequential-order path"
    if case == "image16M":
        assert nseg == 256 and 1 <= fb <= 24


def test_accumulate_helper_is_the_sequential_loop():
    rng = np.random.default_rng(1)
    v = np.ascontiguousarray(rng.integers(0, 195076, 50000), np.uint32)
    assert _seq_sum_fast(0.25, v).view(np.uint32) == _seq_sum(0.25, v).view(np.uint32)
